// mlp_fused_save_hoist.hip -- training-forward instantiations of the fused MLP kernel (every layer's output and LeakyReLU mask
// also written, mlp_kernel.h SAVE) with the per-ray constant terms hoisted (HOIST, objnerf_train_args.ray_bias_ws); a translation
// unit of its own so that it compiles in parallel with mlp_fused.hip / mlp_fused_hoist.hip.
#include "mlp_kernel.h"
#include "host_api.h"

namespace objnerf {

template <bool VOXEL, bool OB>
static void launch(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s, float* save_ws, unsigned* mask_ws) {
  hipLaunchKernelGGL((mlp_kernel<VOXEL, true, true, OB, false, true, true>), dim3(grid), dim3(256), 0, s, a, ntiles, save_ws, mask_ws);
}

int launch_mlp_fused_save_hoist(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s, float* save_ws, unsigned* mask_ws) {
#ifdef OBJ_TUNE_ONLY_MAIN
  return set_error(-9, "tuning build: training kernels are not compiled");
#else
  if (!a.do_scene || !save_ws || !a.ray_bias) return set_error(-1, "mlp_train_forward(fused, hoisted): scene branch, workspace and ray_bias");
  const bool ob = a.do_object != 0;
  if (a.use_voxel) { if (ob) launch<true, true>(a, ntiles, grid, s, save_ws, mask_ws); else launch<true, false>(a, ntiles, grid, s, save_ws, mask_ws); }
  else { if (ob) launch<false, true>(a, ntiles, grid, s, save_ws, mask_ws); else launch<false, false>(a, ntiles, grid, s, save_ws, mask_ws); }
  return check_launch("mlp_train_forward(fused, hoisted)");
#endif
}

}  // namespace objnerf
