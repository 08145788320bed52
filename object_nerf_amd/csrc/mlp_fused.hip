// mlp_fused.hip -- instantiations of the fused (embed-in-registers) MLP kernel (the hoisted ones live in mlp_fused_hoist.hip:
// separate translation units compile in parallel).
#include "mlp_kernel.h"
#include "host_api.h"

namespace objnerf {

template <bool VOXEL, bool SC, bool OB>
static void launch(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s) {
  hipLaunchKernelGGL((mlp_kernel<VOXEL, true, SC, OB>), dim3(grid), dim3(256), 0, s, a, ntiles, nullptr, nullptr);
}
#ifndef OBJ_TUNE_ONLY_MAIN
// density query on points / a lattice (objnerf_mlp_args.points, lat_*): one branch, stops after the sigma head
template <bool VOXEL, bool SC>
static void launch_query(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s) {
  hipLaunchKernelGGL((mlp_kernel<VOXEL, true, SC, !SC, true>), dim3(grid), dim3(256), 0, s, a, ntiles, nullptr, nullptr);
}
// training forward: scene (+ object) branch, every layer's activations also written to save_ws
template <bool VOXEL, bool OB>
static void launch_save(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s, float* save_ws, unsigned* mask_ws) {
  hipLaunchKernelGGL((mlp_kernel<VOXEL, true, true, OB, false, true>), dim3(grid), dim3(256), 0, s, a, ntiles, save_ws, mask_ws);
}
#endif

int launch_mlp_fused(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s, float* save_ws, unsigned* mask_ws) {
  if (a.ray_bias && !save_ws) return launch_mlp_fused_hoist(a, ntiles, grid, s);      // (incl. the hoisted object density query)
  if (a.ray_bias && save_ws) return launch_mlp_fused_save_hoist(a, ntiles, grid, s, save_ws, mask_ws);
  const bool sc = a.do_scene != 0, ob = a.do_object != 0;
#ifdef OBJ_TUNE_ONLY_MAIN
  if (save_ws) return set_error(-9, "tuning build: training kernels are not compiled");
  if (a.sigma_only) return set_error(-9, "tuning build: the density query is not compiled");
#else
  if (a.sigma_only) {
    if (save_ws) return set_error(-1, "mlp_eval(points, sigma_only): inference only");
    if (a.use_voxel) { if (sc) launch_query<true, true>(a, ntiles, grid, s); else launch_query<true, false>(a, ntiles, grid, s); }
    else { if (sc) launch_query<false, true>(a, ntiles, grid, s); else launch_query<false, false>(a, ntiles, grid, s); }
    return check_launch("mlp_eval(points, sigma_only)");
  }
  if (save_ws) {
    if (!sc) return set_error(-1, "mlp_eval(fused, training): the scene branch is always evaluated");
    if (a.use_voxel) { if (ob) launch_save<true, true>(a, ntiles, grid, s, save_ws, mask_ws); else launch_save<true, false>(a, ntiles, grid, s, save_ws, mask_ws); }
    else { if (ob) launch_save<false, true>(a, ntiles, grid, s, save_ws, mask_ws); else launch_save<false, false>(a, ntiles, grid, s, save_ws, mask_ws); }
    return check_launch("mlp_train_forward(fused)");
  }
#endif
#ifdef OBJ_TUNE_ONLY_MAIN   // tuning builds (tools/tune_mlp.py): only the bench instantiation, to compile fast
  if (!(a.use_voxel && sc && ob)) return set_error(-9, "tuning build: only voxel scene+object is compiled");
  launch<true, true, true>(a, ntiles, grid, s);
  return check_launch("mlp_eval(fused)");
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
  return check_launch("mlp_eval(fused)");
#endif
}

}  // namespace objnerf
