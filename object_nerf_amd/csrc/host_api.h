// host_api.h -- helpers shared by the translation units that implement the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/objnerf_hip.h"

namespace objnerf {
// records `msg` as the calling thread's last error and returns `code`
int set_error(int code, const char* msg);
// hipGetLastError() after a launch -> 0 or a recorded negative error
int check_launch(const char* what);

// MLP kernel launchers, one translation unit per input form (compile time)
// save_ws != nullptr: training forward, every layer's activations are also written to the train.hip workspace
// mask_ws: (training forward only) where the LeakyReLU sign masks of the saved activations go (mlp_kernel.h), or nullptr
int launch_mlp_fused(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s, float* save_ws = nullptr,
                     unsigned* mask_ws = nullptr);
int launch_mlp_fused_hoist(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s);                // ray_bias
int launch_mlp_fused_save_hoist(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s, float* save_ws, unsigned* mask_ws);   // ray_bias, training
int launch_mlp_memory(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s, float* save_ws = nullptr,
                      unsigned* mask_ws = nullptr);
// training: fused dgrad chain through the hidden layers (mlp_bwd.hip); act / dz in the workspace layout of train.hip
int launch_mlp_bwd(const float* blob_bwd, const float* aux, long P, const float* act, float* dz, const float* d_sigma,
                   const float* t2, const float* d_isigma, const float* t2i, bool do_object, const unsigned* masks, bool dx,
                   float* d_emb, long ld_emb, float* d_ov, hipStream_t s);
// training: positional-encoding backward + trilinear scatter into the table gradient (train_kernels.hip); emb_xyz / obj_voxel optional
int launch_voxel_embed_bwd(const objnerf_voxel_grid* grid, const float* xyz, long n, const float* d_scene_ftr, const float* d_obj_ftr,
                           float* table_grad, const float* emb_xyz, const float* obj_voxel, hipStream_t s);
// any architecture: a run of plain (32 nt) -> (32 nt) hidden layers in one persistent kernel (chain_generic.hip)
constexpr int kChainMaxLayers = 16;
constexpr int kChainMinWidth = 96;      // three out tiles
int64_t chain_scratch_floats(int width, int layers);
int launch_chain(int width, int L, const float* const* Ws, const float* const* bs, const float* X, long ldx, float* Y, long ldy, long P,
                 int act_last, float* scratch, hipStream_t s);
// any architecture: a whole branch (first / skip layers from 32-column blocks of the input tensors, plain layers chained, density
// head, final layer) in one persistent kernel; returns 1 when the shape is outside what it takes (chain_generic.hip)
struct BranchInput { const float* x; int c; };
int64_t branch_scratch_floats(int width, int D, int nskips, int in_a, int in_b, int in_c, int in_dir);
// returns 0: sigma and fin written; 2: sigma and rgb written (direction layer + colour head in the kernel); 1: shape not taken; < 0 error
int launch_branch(int width, int D, const int32_t* skips, int nskips, const float* const* q, const BranchInput* in, int nin, long P,
                  float* sigma, float* fin, bool sigma_only, const float* emb_dir, int in_dir, float* rgb, float* scratch,
                  hipStream_t s, float* const* saves = nullptr, float* save_dirh = nullptr);
// (saves: training -- D pointers, layer l's output rows are also written to saves[l]; `fin` and save_dirh keep the final and the
// direction layer's rows)
// floats of the mask area behind the activation matrices of a training workspace (mlp_kernel.h: train_mask_floats)
long train_mask_floats_host(long n_points);
// persistent grid of the MLP kernel: one workgroup per CU
unsigned mlp_grid(long ntiles);
}  // namespace objnerf
