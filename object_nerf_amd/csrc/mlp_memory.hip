// mlp_memory.hip -- instantiations of the MLP kernel that reads pre-embedded inputs
// (backs ObjectNeRF.forward / forward_instance, nerf_model.py:97-152, called directly by
// tools/extract_mesh.py:85-108, and the teacher-forced per-branch parity tests).
#include "mlp_kernel.h"
#include "host_api.h"

namespace objnerf {

#ifndef OBJ_TUNE_STUB_MEMORY
template <bool VOXEL, bool SC, bool OB>
static void launch(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s, float* save_ws, unsigned* mask_ws) {
  if (save_ws)      // training forward: also writes every layer's activations
    hipLaunchKernelGGL((mlp_kernel<VOXEL, false, SC, OB, false, true>), dim3(grid), dim3(256), 0, s, a, ntiles, save_ws, mask_ws);
  else if (a.sigma_only)
    hipLaunchKernelGGL((mlp_kernel<VOXEL, false, SC, OB, true>), dim3(grid), dim3(256), 0, s, a, ntiles, nullptr, nullptr);
  else
    hipLaunchKernelGGL((mlp_kernel<VOXEL, false, SC, OB, false>), dim3(grid), dim3(256), 0, s, a, ntiles, nullptr, nullptr);
}
#endif

#ifdef OBJ_TUNE_STUB_MEMORY
int launch_mlp_memory(const objnerf_mlp_args&, long, unsigned, hipStream_t, float*, unsigned*) {
  return set_error(-9, "tuning build: memory-form kernels are not compiled");
}
#else
int launch_mlp_memory(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s, float* save_ws, unsigned* mask_ws) {
  const bool sc = a.do_scene != 0, ob = a.do_object != 0;
  if (sc && ob) return set_error(-1, "mlp_eval(memory): one branch per call (forward or forward_instance)");
  if (save_ws && a.sigma_only) return set_error(-1, "mlp_eval(memory): the training forward needs every layer");
  if (a.use_voxel) {
    if (sc) launch<true, true, false>(a, ntiles, grid, s, save_ws, mask_ws);
    else launch<true, false, true>(a, ntiles, grid, s, save_ws, mask_ws);
  } else {
    if (sc) launch<false, true, false>(a, ntiles, grid, s, save_ws, mask_ws);
    else launch<false, false, true>(a, ntiles, grid, s, save_ws, mask_ws);
  }
  return check_launch("mlp_eval(memory)");
}
#endif

}  // namespace objnerf
