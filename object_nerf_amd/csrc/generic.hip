// generic.hip -- the dual-branch MLP and its embeddings for ANY architecture the reference's `config.model` can describe
// (models/nerf_model.py:18-95: D, W, skips, inst_D, inst_W, inst_skips, N_freq_*, voxel channel counts, code length;
// models/embedding_helper.py:40-74 logscale=False; :77-84 any channel count).
//
// The persistent kernel of mlp_kernel.h is specialised -- register tiling, weight stream layout, hoisting -- for the
// architecture every shipped reference config uses (config/default_conf.yml:7-36).  Other shapes do not get a CPU or PyTorch
// fallback: they run HERE, layer by layer on the fp32 MFMA GEMM of gemm.h (bias / LeakyReLU / sigmoid in the epilogue,
// torch.cat inputs as column blocks accumulated into the same output, never materialised), on the reference's own
// nn.Linear tensors, with the activations in a caller-provided workspace.  Same arithmetic class as the training path's
// layer-wise forward (train.hip); slower than the fused kernel (activations travel through memory) but exact in shape.
#include <hip/hip_runtime.h>
#include <string.h>
#include "layout.h"
#include "device_math.h"
#include "gemm.h"
#include "host_api.h"

namespace objnerf {

int gemm_launch(const GemmArgs& g0, hipStream_t s);      // train.hip

// EmbeddingVoxel.compute_voxel_features_sparse (embedding_helper.py:331-389) before the positional encoding, C channels per
// table row: trilinear weights as the products (a*b)*c, corners in itertools.product order, invalid corners contribute 0.
// One thread per (point, channel); a point's channels sit in neighbouring lanes (one row read per corner).
__global__ void __launch_bounds__(256) voxel_features_kernel(const objnerf_voxel_grid g, int C, const float* __restrict__ xyz,
                                                             long n, float* __restrict__ out, long ldo) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * C) return;
  const long p = idx / C;
  const int c = (int)(idx - p * C);
  const float x = xyz[p * 3], y = xyz[p * 3 + 1], z = xyz[p * 3 + 2];
  const float sx = __fdiv_rn(x + g.offset[0], g.voxel_size);
  const float sy = __fdiv_rn(y + g.offset[1], g.voxel_size);
  const float sz = __fdiv_rn(z + g.offset[2], g.voxel_size);
  const float qx = floorf(sx), qy = floorf(sy), qz = floorf(sz);
  const float u = sx - qx, v = sy - qy, w = sz - qz;
  const float lu = 1.f - u, lv = 1.f - v, lw = 1.f - w;
  float wt[8];
  wt[0] = (lu * lv) * lw; wt[1] = (lu * lv) * w; wt[2] = (lu * v) * lw; wt[3] = (lu * v) * w;
  wt[4] = (u * lv) * lw;  wt[5] = (u * lv) * w;  wt[6] = (u * v) * lw;  wt[7] = (u * v) * w;
  const float X = (float)g.shape[0], Y = (float)g.shape[1], Z = (float)g.shape[2];
  float f = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float cx = qx + (float)((k >> 2) & 1), cy = qy + (float)((k >> 1) & 1), cz = qz + (float)(k & 1);
    const bool ok = cx >= 0.f && cx < X && cy >= 0.f && cy < Y && cz >= 0.f && cz < Z;
    int r = -1;
    if (ok) {
      r = g.idx_map[((size_t)(int)cx * g.shape[1] + (int)cy) * g.shape[2] + (int)cz];
      if (r >= g.n_rows) r = -1;
    }
    const float fv = r < 0 ? 0.f : g.table[(size_t)r * C + c];
    f = k == 0 ? fv * wt[k] : f + fv * wt[k];
  }
  out[p * ldo + c] = f;
}

// Embedding.forward (embedding_helper.py:57-74) on a column block: x (n, C) with row stride ldx -> out (n, C*(2F+1)) with row
// stride ldo -- every embedding of a torch.cat([...], -1) is written straight into its columns of the concatenated matrix
__global__ void pos_encode_block_kernel(const float* __restrict__ x, long ldx, long n, int C, int F,
                                        const float* __restrict__ freqs, float* __restrict__ out, long ldo) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * C) return;
  const long row = idx / C;
  const int c = (int)(idx - row * C);
  const float v = x[row * ldx + c];
  float* o = out + row * ldo;
  o[c] = v;
  float f = 1.f;
  for (int k = 0; k < F; ++k) {
    const SinCos sc = psincos<true>((freqs ? freqs[k] : f) * v);
    o[C * (1 + 2 * k) + c] = sc.s;
    o[C * (2 + 2 * k) + c] = sc.c;
    f *= 2.f;
  }
}

// out[r*ldo + c] = src[(r / repeat)*lds + c]: a per-ray row repeated over the ray's samples (rendering.py:89-94) as a column block
__global__ void repeat_rows_kernel(const float* __restrict__ src, long lds, long n_rows, int C, int repeat, float* __restrict__ out,
                                   long ldo) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * C) return;
  const long r = idx / C;
  const int c = (int)(idx - r * C);
  out[r * ldo + c] = src[(r / repeat) * lds + c];
}

struct GCtx { hipStream_t s; int rc; };
static void lin(GCtx& c, const float* X, long ldx, const float* W, long ldw, long P, int out, int in, float* Y, long ldy,
                int accumulate, int epi, const float* bias) {
  if (c.rc) return;
  GemmArgs g{X, ldx, 1, W, ldw, 1, Y, ldy, P, out, in, accumulate, epi, bias, 1, nullptr};
  c.rc = gemm_launch(g, c.s);
}
static bool has(const int32_t* list, int n, int v) {
  for (int i = 0; i < n; ++i) if (list[i] == v) return true;
  return false;
}
static inline unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }

}  // namespace objnerf

using namespace objnerf;

extern "C" {

int objnerf_voxel_features(const objnerf_voxel_grid* grid, int C, const float* xyz, int64_t n, float* out, int64_t ldo,
                           void* stream) {
  if (!grid || !grid->idx_map || !grid->table || !xyz || !out || C < 1 || ldo < C) return set_error(-1, "voxel_features: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(voxel_features_kernel, dim3(nblk(n * C)), dim3(256), 0, (hipStream_t)stream, *grid, C, xyz, (long)n, out, (long)ldo);
  return check_launch("voxel_features");
}

int objnerf_pos_encode_block(const float* x, int64_t ldx, int64_t n, int C, int n_freqs, const float* freqs, float* out,
                             int64_t ldo, void* stream) {
  if (!x || !out || C < 1 || n_freqs < 0 || ldx < C || ldo < (int64_t)C * (2 * n_freqs + 1))
    return set_error(-1, "pos_encode_block: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(pos_encode_block_kernel, dim3(nblk(n * C)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (long)n, C,
                     n_freqs, freqs, out, (long)ldo);
  return check_launch("pos_encode_block");
}

int objnerf_repeat_rows(const float* src, int64_t lds, int64_t n_rows, int C, int repeat, float* out, int64_t ldo, void* stream) {
  if (!src || !out || C < 1 || repeat < 1 || lds < C || ldo < C) return set_error(-1, "repeat_rows: bad arguments");
  if (n_rows == 0) return 0;
  hipLaunchKernelGGL(repeat_rows_kernel, dim3(nblk(n_rows * C)), dim3(256), 0, (hipStream_t)stream, src, (long)lds, (long)n_rows, C,
                     repeat, out, (long)ldo);
  return check_launch("repeat_rows");
}

int objnerf_arch_num_param_ptrs(const objnerf_arch* a) { return a ? 2 * (a->D + 4) + 2 * (a->inst_D + 4) : -1; }

static int check_arch(const objnerf_arch* a) {
  if (!a || a->D < 1 || a->D > 64 || a->W < 2 || (a->W & 1) || a->inst_D < 1 || a->inst_D > 64 || a->inst_W < 2 || (a->inst_W & 1) ||
      a->n_skips < 0 || a->n_skips > 8 || a->n_inst_skips < 0 || a->n_inst_skips > 8 || a->in_xyz < 1 || a->in_dir < 1 ||
      a->obj_voxel_c < 0 || a->code_c < 0)
    return set_error(-1, "mlp_generic: bad architecture (1 <= D, inst_D <= 64; even W, inst_W; at most 8 skips per branch)");
  return 0;
}

int64_t objnerf_mlp_generic_workspace_floats(const objnerf_arch* a, int64_t n_points) {
  if (check_arch(a) || n_points < 0) return -1;
  const int64_t wmax = a->W > a->inst_W ? a->W : a->inst_W;
  return n_points * (2 * wmax + wmax + wmax / 2);          // two ping-pong hidden buffers | final | direction hidden
}

int objnerf_mlp_generic(const objnerf_mlp_generic_args* g, void* stream) {
  if (!g || !g->h_params || !g->emb_xyz || !g->workspace) return set_error(-1, "mlp_generic: bad arguments");
  const objnerf_arch* a = &g->arch;
  if (check_arch(a)) return -1;
  if (!g->do_scene && !g->do_object) return set_error(-1, "mlp_generic: no branch selected");
  if (!g->sigma_only && !g->emb_dir) return set_error(-1, "mlp_generic: emb_dir is needed for the colour layers");
  if (g->do_scene && (!g->sigma || (!g->sigma_only && !g->rgb))) return set_error(-1, "mlp_generic: scene outputs missing");
  if (g->do_object && (!g->inst_sigma || (!g->sigma_only && !g->inst_rgb) || (a->code_c > 0 && !g->obj_code) ||
                       (a->obj_voxel_c > 0 && !g->obj_voxel)))
    return set_error(-1, "mlp_generic: object branch inputs / outputs missing");
  const long P = g->n_points;
  if (P < 0) return set_error(-1, "mlp_generic: negative n_points");
  if (P == 0) return 0;
  const float* const* p = g->h_params;
  const int npar = objnerf_arch_num_param_ptrs(a);
  for (int i = 0; i < npar; ++i) {
    const bool scene = i < 2 * (a->D + 4);
    if ((scene ? g->do_scene : g->do_object) && !p[i]) return set_error(-1, "mlp_generic: null parameter pointer");
  }
  GCtx c{(hipStream_t)stream, 0};
  const long wmax = a->W > a->inst_W ? a->W : a->inst_W;
  float* buf[2] = {g->workspace, g->workspace + P * wmax};
  float* fin = g->workspace + 2 * P * wmax;
  float* dirh = fin + P * wmax;

  // one branch: layers l = 0 .. D-1 (LeakyReLU; layer l in `skips` sees cat([input, h]), nerf_model.py:104-105, 137-138),
  // sigma head (no activation), final (no activation), direction layer cat([final, emb_dir]) -> W/2 LeakyReLU, rgb head
  // sigmoid.  The input is itself a cat of up to three column blocks (object branch: emb_xyz | obj_voxel | obj_code).
  struct Blk { const float* x; int c; };
  auto branch = [&](const float* const* q, int D, int W, const int32_t* skips, int nsk, const Blk* in, int nin, float* sig, float* rgb) {
    int cin = 0;
    for (int i = 0; i < nin; ++i) cin += in[i].c;
    // y (+)= cat(in) * W[:, col0 : col0 + cin]^T, epilogue on the last block unless more follows
    auto input_blocks = [&](const float* Wm, long ldw, int col0, float* y, int acc_first, int epi_last, const float* bias) {
      int col = col0;
      for (int i = 0; i < nin; ++i) {
        const bool last = i == nin - 1;
        lin(c, in[i].x, in[i].c, Wm + col, ldw, P, W, in[i].c, y, W, (i == 0 ? acc_first : 1), last ? epi_last : EPI_NONE, last ? bias : nullptr);
        col += in[i].c;
      }
    };
    const float* h = nullptr;
    for (int l = 0; l < D; ++l) {
      float* y = buf[l & 1];
      const float* Wm = q[2 * l];
      const float* b = q[2 * l + 1];
      if (l == 0) {
        input_blocks(Wm, cin, 0, y, 0, EPI_BIAS_LEAKY, b);
      } else if (has(skips, nsk, l)) {
        // hidden block first (columns cin .. cin + W), then the input blocks with the epilogue on the last one
        lin(c, h, W, Wm + cin, cin + W, P, W, W, y, W, 0, EPI_NONE, nullptr);
        input_blocks(Wm, cin + W, 0, y, 1, EPI_BIAS_LEAKY, b);
      } else {
        lin(c, h, W, Wm, W, P, W, W, y, W, 0, EPI_BIAS_LEAKY, b);
      }
      h = y;
    }
    const float* const* t = q + 2 * D;          // final, dir, sigma, rgb
    lin(c, h, W, t[4], W, P, 1, W, sig, 1, 0, EPI_BIAS, t[5]);
    if (g->sigma_only) return;
    lin(c, h, W, t[0], W, P, W, W, fin, W, 0, EPI_BIAS, t[1]);
    lin(c, fin, W, t[2], W + a->in_dir, P, W / 2, W, dirh, W / 2, 0, EPI_NONE, nullptr);
    lin(c, g->emb_dir, a->in_dir, t[2] + W, W + a->in_dir, P, W / 2, a->in_dir, dirh, W / 2, 1, EPI_BIAS_LEAKY, t[3]);
    lin(c, dirh, W / 2, t[6], W / 2, P, 3, W / 2, rgb, 3, 0, EPI_BIAS_SIGMOID, t[7]);
  };
  if (g->do_scene) {
    const Blk in[1] = {{g->emb_xyz, a->in_xyz}};
    branch(p, a->D, a->W, a->skips, a->n_skips, in, 1, g->sigma, g->rgb);
  }
  if (g->do_object) {
    Blk in[3];
    int nin = 0;
    in[nin++] = {g->emb_xyz, a->in_xyz};
    if (a->obj_voxel_c > 0) in[nin++] = {g->obj_voxel, a->obj_voxel_c};
    if (a->code_c > 0) in[nin++] = {g->obj_code, a->code_c};
    branch(p + 2 * (a->D + 4), a->inst_D, a->inst_W, a->inst_skips, a->n_inst_skips, in, nin, g->inst_sigma, g->inst_rgb);
  }
  return c.rc;
}

}  // extern "C"
