// mlp_fused_hoist.hip -- inference instantiations of the fused MLP kernel with the per-ray constant terms
// hoisted (mlp_kernel.h HOIST, objnerf_mlp_args.ray_bias); a translation unit of its own so that it compiles in parallel
// with mlp_fused.hip.
#include "mlp_kernel.h"
#include "host_api.h"

namespace objnerf {

template <bool VOXEL, bool SC, bool OB>
static void launch(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s) {
  hipLaunchKernelGGL((mlp_kernel<VOXEL, true, SC, OB, false, false, true>), dim3(grid), dim3(256), 0, s, a, ntiles, nullptr, nullptr);
}

int launch_mlp_fused_hoist(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s) {
  const bool sc = a.do_scene != 0, ob = a.do_object != 0;
#ifndef OBJ_TUNE_ONLY_MAIN
  if (a.sigma_only) {        // object-branch density query on points / a lattice with the code's terms hoisted
    if (sc || !ob) return set_error(-1, "mlp_eval(points, sigma_only): ray_bias serves the object query only");
    if (a.use_voxel) hipLaunchKernelGGL((mlp_kernel<true, true, false, true, true, false, true>), dim3(grid), dim3(256), 0, s, a, ntiles, nullptr, nullptr);
    else hipLaunchKernelGGL((mlp_kernel<false, true, false, true, true, false, true>), dim3(grid), dim3(256), 0, s, a, ntiles, nullptr, nullptr);
    return check_launch("mlp_eval(points, sigma_only, hoisted)");
  }
#endif
#ifdef OBJ_TUNE_ONLY_MAIN   // tuning builds (tools/tune_mlp.py): only the bench instantiation, to compile fast
  if (!(a.use_voxel && sc && ob)) return set_error(-9, "tuning build: only voxel scene+object is compiled");
  launch<true, true, true>(a, ntiles, grid, s);
#else
  if (a.use_voxel) {
    if (sc && ob) launch<true, true, true>(a, ntiles, grid, s);
    else if (sc) launch<true, true, false>(a, ntiles, grid, s);
    else launch<true, false, true>(a, ntiles, grid, s);
  } else {
    if (sc && ob) launch<false, true, true>(a, ntiles, grid, s);
    else if (sc) launch<false, true, false>(a, ntiles, grid, s);
    else launch<false, false, true>(a, ntiles, grid, s);
  }
#endif
  return check_launch("mlp_eval(fused, hoisted)");
}

}  // namespace objnerf
