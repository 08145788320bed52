// mlp_fused_b3.hip -- split-bf16 instantiations (objnerf_mlp_args.mfma_bf16x3) of the fused MLP kernel: inference and
// the training forward.  Same dispatch as mlp_fused.hip.
#include "mlp_kernel.h"
#include "host_api.h"

namespace objnerf {

template <bool VOXEL, bool SC, bool OB>
static void launch(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s) {
  hipLaunchKernelGGL((mlp_kernel<VOXEL, true, SC, OB, false, false, true>), dim3(grid), dim3(256), 0, s, a, ntiles, nullptr);
}
#ifndef OBJ_TUNE_ONLY_MAIN
// training forward: scene (+ object) branch, every layer's activations also written to save_ws
template <bool VOXEL, bool OB>
static void launch_save(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s, float* save_ws) {
  hipLaunchKernelGGL((mlp_kernel<VOXEL, true, true, OB, false, true, true>), dim3(grid), dim3(256), 0, s, a, ntiles, save_ws);
}
#endif

// memory form (pre-embedded inputs: ObjectNeRF.forward / forward_instance) in the split-bf16 mode: the teacher-forced
// per-branch parity test grades the mode on it at the fp32 kernel's tolerance
template <bool VOXEL, bool SC, bool OB>
static void launch_mem(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s) {
  hipLaunchKernelGGL((mlp_kernel<VOXEL, false, SC, OB, false, false, true>), dim3(grid), dim3(256), 0, s, a, ntiles, nullptr);
}
int launch_mlp_memory_b3(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s) {
#ifdef OBJ_TUNE_ONLY_MAIN
  return set_error(-9, "tuning build: memory-form kernels are not compiled");
#else
  const bool sc = a.do_scene != 0;
  if (a.use_voxel) { if (sc) launch_mem<true, true, false>(a, ntiles, grid, s); else launch_mem<true, false, true>(a, ntiles, grid, s); }
  else { if (sc) launch_mem<false, true, false>(a, ntiles, grid, s); else launch_mem<false, false, true>(a, ntiles, grid, s); }
  return check_launch("mlp_eval(memory, split-bf16)");
#endif
}

int launch_mlp_fused_b3(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s, float* save_ws) {
  if (a.ray_bias && !save_ws) return launch_mlp_fused_b3_hoist(a, ntiles, grid, s);
  const bool sc = a.do_scene != 0, ob = a.do_object != 0;
#ifdef OBJ_TUNE_ONLY_MAIN
  if (save_ws) return set_error(-9, "tuning build: training kernels are not compiled");
#else
  if (save_ws) {
    if (!sc) return set_error(-1, "mlp_eval(fused, training): the scene branch is always evaluated");
    if (a.use_voxel) { if (ob) launch_save<true, true>(a, ntiles, grid, s, save_ws); else launch_save<true, false>(a, ntiles, grid, s, save_ws); }
    else { if (ob) launch_save<false, true>(a, ntiles, grid, s, save_ws); else launch_save<false, false>(a, ntiles, grid, s, save_ws); }
    return check_launch("mlp_train_forward(fused, split-bf16)");
  }
#endif
#ifdef OBJ_TUNE_ONLY_MAIN   // tuning builds (tools/tune_mlp.py): only the bench instantiation, to compile fast
  if (!(a.use_voxel && sc && ob)) return set_error(-9, "tuning build: only voxel scene+object is compiled");
  launch<true, true, true>(a, ntiles, grid, s);
  return check_launch("mlp_eval(fused, split-bf16)");
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
  return check_launch("mlp_eval(fused, split-bf16)");
#endif
}

}  // namespace objnerf
