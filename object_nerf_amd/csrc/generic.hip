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

// Backward of pos_encode_block w.r.t. its input: d_x[c] (+)= d_out[c] + sum_k f_k (cos(f_k x) d_sin_k - sin(f_k x) d_cos_k)
__global__ void pos_encode_block_bwd_kernel(const float* __restrict__ x, long ldx, long n, int C, int F, const float* __restrict__ freqs,
                                            const float* __restrict__ d_out, long ldo, float* __restrict__ d_x, long ldd) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * C) return;
  const long row = idx / C;
  const int c = (int)(idx - row * C);
  const float v = x[row * ldx + c];
  const float* o = d_out + row * ldo;
  float g = o[c];
  float f = 1.f;
  for (int k = 0; k < F; ++k) {
    const float fk = freqs ? freqs[k] : f;
    const SinCos sc = psincos<true>(fk * v);
    g += fk * (sc.c * o[C * (1 + 2 * k) + c] - sc.s * o[C * (2 + 2 * k) + c]);
    f *= 2.f;
  }
  d_x[row * ldd + c] = g;
}

// Backward of voxel_features_kernel w.r.t. the table: table_grad[row of corner k][c] += w_k * d_raw[p][c] (fp32 atomics; the
// default architecture's voxel_embed_backward kernel pre-sums a workgroup's contributions in LDS -- this generic form does not)
__global__ void __launch_bounds__(256) voxel_features_bwd_kernel(const objnerf_voxel_grid g, int C, const float* __restrict__ xyz, long n,
                                                                 const float* __restrict__ d_raw, long ldd, float* __restrict__ table_grad) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * C) return;
  const long p = idx / C;
  const int c = (int)(idx - p * C);
  const float dv = d_raw[p * ldd + c];
  if (dv == 0.f) return;
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
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float cx = qx + (float)((k >> 2) & 1), cy = qy + (float)((k >> 1) & 1), cz = qz + (float)(k & 1);
    if (!(cx >= 0.f && cx < X && cy >= 0.f && cy < Y && cz >= 0.f && cz < Z)) continue;
    const int r = g.idx_map[((size_t)(int)cx * g.shape[1] + (int)cy) * g.shape[2] + (int)cz];
    if (r < 0 || r >= g.n_rows) continue;
    atomicAdd(table_grad + (size_t)r * C + c, wt[k] * dv);
  }
}

__global__ void sigmoid_bwd_generic_kernel(float* __restrict__ dz, const float* __restrict__ dy, const float* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dz[i] = dy[i] * (y[i] * (1.f - y[i]));
}

struct GCtx { hipStream_t s; int rc; };
// dX (P x in) (+)= dY (P x out) * W (out x in); act != null: the LeakyReLU backward of the layer dX belongs to, on the complete sum
static void dgrad(GCtx& c, const float* dY, long lddy, const float* W, long ldw, long P, int out, int in, float* dX, long lddx,
                  int accumulate, const float* act = nullptr) {
  if (c.rc) return;
  GemmArgs g{dY, lddy, 1, W, ldw, 0, dX, lddx, P, in, out, accumulate, act ? EPI_LEAKY_BWD : EPI_NONE, act, 1, nullptr};
  c.rc = gemm_launch(g, c.s);
}
// dW (out x in, ld ldw) += dY^T X, split over the points with fp32 atomics; db: the bias gradient from the same tiles
static void wgrad(GCtx& c, const float* dY, long lddy, const float* X, long ldx, long P, int out, int in, float* dW, long ldw,
                  float* db = nullptr) {
  if (c.rc) return;
  const long tiles = ((out + GBM - 1) / GBM) * (long)((in + GBN - 1) / GBN);
  long split = P / (tiles >= 2 ? 1024 : 512);
  const long max_split = (P + 4 * GBK - 1) / (4 * GBK);
  if (split > max_split) split = max_split;
  if (split < 2) split = 2;                       // >= 2 forces the atomic += path (dW always accumulates)
  GemmArgs g{dY, lddy, 0, X, ldx, 0, dW, ldw, out, in, P, 1, EPI_NONE, nullptr, (int)split, db};
  c.rc = gemm_launch(g, c.s);
}
static void lin(GCtx& c, const float* X, long ldx, const float* W, long ldw, long P, int out, int in, float* Y, long ldy,
                int accumulate, int epi, const float* bias) {
  if (c.rc) return;
  GemmArgs g{X, ldx, 1, W, ldw, 1, Y, ldy, P, out, in, accumulate, epi, bias, 1, nullptr};
  c.rc = gemm_launch(g, c.s);
}
// y = act(cat(x_0 .. x_{n-1}) * W^T + b): the column blocks of a torch.cat input as ONE segmented contraction (gemm.h: up to four
// segments, each with its own operand pointers) instead of one accumulating launch per block -- no read-modify-write passes over
// y, and the 27- / 64- / 104-column blocks ride inside a kernel that is already running instead of costing a launch and two trips
// of y through memory each (round 5: the layer-wise path was 28 launches per MLP call, 8 of them such accumulations).
// OBJNERF_GENERIC_CAT=launches restores the per-block launches (developer A/B switch).
struct CatBlk { const float* x; long ldx; int c; const float* W; };          // W: first weight column of the block (row stride ldw)
static void lin_cat(GCtx& c, const CatBlk* blk, int nblk_, long ldw, long P, int out, float* Y, long ldy, int epi, const float* bias) {
  if (c.rc || nblk_ <= 0) return;
  static const bool per_block = [] { const char* e = getenv("OBJNERF_GENERIC_CAT"); return e && !strcmp(e, "launches"); }();
  if (per_block || nblk_ > 4) {
    for (int i = 0; i < nblk_; ++i) {
      const bool last = i == nblk_ - 1;
      lin(c, blk[i].x, blk[i].ldx, blk[i].W, ldw, P, out, blk[i].c, Y, ldy, i == 0 ? 0 : 1, last ? epi : EPI_NONE, last ? bias : nullptr);
    }
    return;
  }
  GemmArgs g{blk[0].x, blk[0].ldx, 1, blk[0].W, ldw, 1, Y, ldy, P, out, blk[0].c, 0, epi, bias, 1, nullptr};
  if (nblk_ > 1) {
    g.nseg = nblk_;
    for (int i = 0; i < nblk_; ++i) {
      g.segA[i] = blk[i].x; g.seglda[i] = blk[i].ldx;
      g.segB[i] = blk[i].W; g.segldb[i] = ldw;
      g.segK[i] = blk[i].c;
    }
  }
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
  // two ping-pong hidden buffers | final | direction hidden | the packed weight stream of a run of plain layers (chain_generic.hip)
  int64_t chain = chain_scratch_floats(a->W, a->D) > chain_scratch_floats(a->inst_W, a->inst_D) ? chain_scratch_floats(a->W, a->D)
                                                                                                  : chain_scratch_floats(a->inst_W, a->inst_D);
  // ... or of a whole branch (first / skip layers as input blocks + plain layers + final + direction layer)
  const int64_t br_s = branch_scratch_floats(a->W, a->D, a->n_skips, a->in_xyz, 0, 0, a->in_dir);
  const int64_t br_o = branch_scratch_floats(a->inst_W, a->inst_D, a->n_inst_skips, a->in_xyz, a->obj_voxel_c, a->code_c, a->in_dir);
  if (br_s > chain) chain = br_s;
  if (br_o > chain) chain = br_o;
  return n_points * (2 * wmax + wmax + wmax / 2) + chain + 4;
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
  float* chain_ws = dirh + P * (wmax / 2);
  chain_ws += (4 - ((chain_ws - g->workspace) & 3)) & 3;           // 16-byte aligned behind the point buffers (the stream is DMA'd)
  // Runs of plain hidden layers (not the first, not in `skips`) of a width that is a multiple of 32 from 96 to 256 go through ONE
  // persistent kernel each (chain_generic.hip, round 6): rows read once, weights from the LDS ring, the layers chained in registers.
  // OBJNERF_GENERIC_CHAIN=0: every layer its own GEMM as in rounds 4-5 (read on every call: the tests switch it inside a process).
  // 3: whole branches incl. the direction layer and the colour head; 2: up to `final`; 1: runs of plain layers only; 0: GEMMs
  const int chain_mode = [] { const char* e = getenv("OBJNERF_GENERIC_CHAIN"); return e ? atoi(e) : 3; }();
  const bool chain_on = chain_mode != 0;

  // one branch: layers l = 0 .. D-1 (LeakyReLU; layer l in `skips` sees cat([input, h]), nerf_model.py:104-105, 137-138),
  // sigma head (no activation), final (no activation), direction layer cat([final, emb_dir]) -> W/2 LeakyReLU, rgb head
  // sigmoid.  The input is itself a cat of up to three column blocks (object branch: emb_xyz | obj_voxel | obj_code).
  struct Blk { const float* x; int c; };
  auto branch = [&](const float* const* q, int D, int W, const int32_t* skips, int nsk, const Blk* in, int nin, float* sig, float* rgb) {
    int cin = 0;
    for (int i = 0; i < nin; ++i) cin += in[i].c;
    // y = act(cat([hidden,] in) * W^T + b) as one segmented product (lin_cat): the hidden block of a skip layer sits BEHIND the
    // input blocks in the weight (columns cin .. cin + W, nerf_model.py:104-105) and is contracted first
    auto cat_layer = [&](const float* Wm, long ldw, const float* hid, float* y, const float* bias) {
      CatBlk blk[5];
      int nb = 0, col = 0;
      if (hid) blk[nb++] = CatBlk{hid, W, W, Wm + cin};
      for (int i = 0; i < nin; ++i) { blk[nb++] = CatBlk{in[i].x, in[i].c, in[i].c, Wm + col}; col += in[i].c; }
      lin_cat(c, blk, nb, ldw, P, W, y, W, EPI_BIAS_LEAKY, bias);
    };
    const bool chain_w = chain_on && W >= kChainMinWidth && W <= 256 && (W & 31) == 0;
    // the whole branch up to `final` in one persistent kernel when the shape allows it (chain_generic.hip: launch_branch)
    bool branch_done = false;
    if (chain_mode >= 2 && W >= 32 && W <= 256 && (W & 31) == 0 && !c.rc) {      // (32 / 64 wide: zero-padded to three out tiles)
      BranchInput bin[3];
      for (int i = 0; i < nin; ++i) bin[i] = BranchInput{in[i].x, in[i].c};
      const int rc = launch_branch(W, D, skips, nsk, q, bin, nin, P, sig, fin, g->sigma_only != 0, chain_mode >= 3 ? g->emb_dir : nullptr,
                                   a->in_dir, rgb, chain_ws, c.s);
      if (rc < 0) c.rc = rc;
      branch_done = rc == 0 || rc == 2;
      if (rc == 2) return;                        // the direction layer and the colour head ran in the kernel as well
    }
    const float* h = nullptr;
    for (int l = 0; l < D && !branch_done; ++l) {
      float* y = buf[l & 1];
      const float* Wm = q[2 * l];
      const float* b = q[2 * l + 1];
      if (l == 0) cat_layer(Wm, cin, nullptr, y, b);
      else if (has(skips, nsk, l)) cat_layer(Wm, cin + W, h, y, b);
      else if (chain_w) {
        int l1 = l;                               // the run of plain layers l .. l1
        while (l1 + 1 < D && !has(skips, nsk, l1 + 1) && l1 + 1 - l < kChainMaxLayers) ++l1;
        const float* Ws[kChainMaxLayers];
        const float* bs[kChainMaxLayers];
        for (int i = l; i <= l1; ++i) { Ws[i - l] = q[2 * i]; bs[i - l] = q[2 * i + 1]; }
        y = buf[l1 & 1];
        if (!c.rc) c.rc = launch_chain(W, l1 - l + 1, Ws, bs, h, W, y, W, P, 1, chain_ws, c.s);
        l = l1;
      } else lin(c, h, W, Wm, W, P, W, W, y, W, 0, EPI_BIAS_LEAKY, b);
      h = y;
    }
    const float* const* t = q + 2 * D;          // final, dir, sigma, rgb
    if (!branch_done) lin(c, h, W, t[4], W, P, 1, W, sig, 1, 0, EPI_BIAS, t[5]);
    if (g->sigma_only) return;
    if (branch_done) {}
    else if (chain_w) { if (!c.rc) c.rc = launch_chain(W, 1, &t[0], &t[1], h, W, fin, W, P, 0, chain_ws, c.s); }     // final: no activation
    else lin(c, h, W, t[0], W, P, W, W, fin, W, 0, EPI_BIAS, t[1]);
    const CatBlk dblk[2] = {{fin, W, W, t[2]}, {g->emb_dir, a->in_dir, a->in_dir, t[2] + W}};       // cat([final, emb_dir])
    lin_cat(c, dblk, 2, W + a->in_dir, P, W / 2, dirh, W / 2, EPI_BIAS_LEAKY, t[3]);
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


// ---- training of such architectures (what loss.backward() does through models/nerf_model.py:97-152): the same layer-wise
// GEMMs with the activations of every layer kept, and their backward (dgrad with the LeakyReLU backward in the epilogue,
// split-K weight gradients with fp32 atomics, gradients w.r.t. the embedding blocks).  The default architecture trains on the
// fused kernels (train.hip, mlp_bwd.hip, wgrad.hip); this is the path every OTHER shape takes.
namespace {
struct GenWs {          // saved activations / their gradients, floats per point: D*W | W | W/2 | inst_D*IW | IW | IW/2
  const objnerf_arch* a; long P; float* base;
  float* A(int l) const { return base + (long)l * a->W * P; }                       // scene layer l = 0 .. D-1
  float* fin() const { return base + (long)a->D * a->W * P; }
  float* dirh() const { return fin() + (long)a->W * P; }
  float* B(int l) const { return dirh() + (long)(a->W / 2) * P + (long)l * a->inst_W * P; }
  float* ofin() const { return B(a->inst_D); }
  float* odirh() const { return ofin() + (long)a->inst_W * P; }
};
long gen_ws_floats(const objnerf_arch* a, long P) {
  return P * ((long)a->D * a->W + a->W + a->W / 2 + (long)a->inst_D * a->inst_W + a->inst_W + a->inst_W / 2);
}
}  // namespace

static int64_t gen_branch_scratch(const objnerf_arch* a) {
  const int64_t br_s = branch_scratch_floats(a->W, a->D, a->n_skips, a->in_xyz, 0, 0, a->in_dir);
  const int64_t br_o = branch_scratch_floats(a->inst_W, a->inst_D, a->n_inst_skips, a->in_xyz, a->obj_voxel_c, a->code_c, a->in_dir);
  return (br_s > br_o ? br_s : br_o) + 4;
}
int64_t objnerf_mlp_generic_train_workspace_floats(const objnerf_arch* a, int64_t n_points) {
  if (check_arch(a) || n_points < 0) return -1;
  // the saved activations, then (round 6) the packed weight stream of a branch for the forward's persistent kernel (chain_generic.hip)
  return gen_ws_floats(a, n_points) + gen_branch_scratch(a);
}
/* backward scratch: the gradients w.r.t. every layer's pre-activation output (same layout) + (P,3) x 2 for the rgb heads */
int64_t objnerf_mlp_generic_train_scratch_floats(const objnerf_arch* a, int64_t n_points) {
  if (check_arch(a) || n_points < 0) return -1;
  return gen_ws_floats(a, n_points) + 6 * n_points;
}

int objnerf_mlp_generic_train_forward(const objnerf_mlp_generic_args* g, void* stream) {
  if (!g || !g->h_params || !g->emb_xyz || !g->emb_dir || !g->workspace || !g->sigma || !g->rgb)
    return set_error(-1, "mlp_generic_train_forward: bad arguments");
  const objnerf_arch* a = &g->arch;
  if (check_arch(a)) return -1;
  if (!g->do_scene || g->sigma_only) return set_error(-1, "mlp_generic_train_forward: the scene branch and every layer are evaluated");
  if (g->do_object && (!g->inst_sigma || !g->inst_rgb || (a->code_c > 0 && !g->obj_code) || (a->obj_voxel_c > 0 && !g->obj_voxel)))
    return set_error(-1, "mlp_generic_train_forward: object branch inputs / outputs missing");
  const long P = g->n_points;
  if (P <= 0) return P < 0 ? set_error(-1, "mlp_generic_train_forward: negative n_points") : 0;
  GCtx c{(hipStream_t)stream, 0};
  const GenWs w{a, P, g->workspace};
  const float* const* p = g->h_params;
  struct Blk { const float* x; int c; };
  // the forward's persistent branch kernel keeps every layer's rows as it goes (chain_generic.hip; OBJNERF_GENERIC_CHAIN < 2: GEMMs)
  const int chain_mode = [] { const char* e = getenv("OBJNERF_GENERIC_CHAIN"); return e ? atoi(e) : 3; }();
  float* chain_ws = g->workspace + gen_ws_floats(a, P);
  chain_ws += (4 - ((chain_ws - g->workspace) & 3)) & 3;
  auto branch = [&](const float* const* q, int D, int W, const int32_t* skips, int nsk, const Blk* in, int nin, auto act_of, float* fin,
                    float* dirh, float* sig, float* rgb) {
    int cin = 0;
    for (int i = 0; i < nin; ++i) cin += in[i].c;
    const float* const* t = q + 2 * D;
    if (chain_mode >= 2 && D <= 64 && !c.rc) {
      BranchInput bin[3];
      for (int i = 0; i < nin; ++i) bin[i] = BranchInput{in[i].x, in[i].c};
      float* saves[64];
      for (int l = 0; l < D; ++l) saves[l] = act_of(l);
      const int rc = launch_branch(W, D, skips, nsk, q, bin, nin, P, sig, fin, false, chain_mode >= 3 ? g->emb_dir : nullptr, a->in_dir, rgb,
                                   chain_ws, c.s, saves, dirh);
      if (rc < 0) { c.rc = rc; return; }
      if (rc == 2) return;                       // every layer, both heads
      if (rc == 0) {                             // up to `final`: the direction layer and the colour head as GEMMs
        const CatBlk dblk[2] = {{fin, W, W, t[2]}, {g->emb_dir, a->in_dir, a->in_dir, t[2] + W}};
        lin_cat(c, dblk, 2, W + a->in_dir, P, W / 2, dirh, W / 2, EPI_BIAS_LEAKY, t[3]);
        lin(c, dirh, W / 2, t[6], W / 2, P, 3, W / 2, rgb, 3, 0, EPI_BIAS_SIGMOID, t[7]);
        return;
      }
    }
    auto cat_layer = [&](const float* Wm, long ldw, const float* hid, float* y, const float* bias) {      // as in objnerf_mlp_generic
      CatBlk blk[5];
      int nb = 0, col = 0;
      if (hid) blk[nb++] = CatBlk{hid, W, W, Wm + cin};
      for (int i = 0; i < nin; ++i) { blk[nb++] = CatBlk{in[i].x, in[i].c, in[i].c, Wm + col}; col += in[i].c; }
      lin_cat(c, blk, nb, ldw, P, W, y, W, EPI_BIAS_LEAKY, bias);
    };
    for (int l = 0; l < D; ++l) {
      float* y = act_of(l);
      if (l == 0) cat_layer(q[0], cin, nullptr, y, q[1]);
      else if (has(skips, nsk, l)) cat_layer(q[2 * l], cin + W, act_of(l - 1), y, q[2 * l + 1]);
      else lin(c, act_of(l - 1), W, q[2 * l], W, P, W, W, y, W, 0, EPI_BIAS_LEAKY, q[2 * l + 1]);
    }
    const float* h = act_of(D - 1);
    lin(c, h, W, t[4], W, P, 1, W, sig, 1, 0, EPI_BIAS, t[5]);
    lin(c, h, W, t[0], W, P, W, W, fin, W, 0, EPI_BIAS, t[1]);
    const CatBlk dblk[2] = {{fin, W, W, t[2]}, {g->emb_dir, a->in_dir, a->in_dir, t[2] + W}};
    lin_cat(c, dblk, 2, W + a->in_dir, P, W / 2, dirh, W / 2, EPI_BIAS_LEAKY, t[3]);
    lin(c, dirh, W / 2, t[6], W / 2, P, 3, W / 2, rgb, 3, 0, EPI_BIAS_SIGMOID, t[7]);
  };
  {
    const Blk in[1] = {{g->emb_xyz, a->in_xyz}};
    branch(p, a->D, a->W, a->skips, a->n_skips, in, 1, [&](int l) { return w.A(l); }, w.fin(), w.dirh(), g->sigma, g->rgb);
  }
  if (g->do_object) {
    Blk in[3];
    int nin = 0;
    in[nin++] = {g->emb_xyz, a->in_xyz};
    if (a->obj_voxel_c > 0) in[nin++] = {g->obj_voxel, a->obj_voxel_c};
    if (a->code_c > 0) in[nin++] = {g->obj_code, a->code_c};
    branch(p + 2 * (a->D + 4), a->inst_D, a->inst_W, a->inst_skips, a->n_inst_skips, in, nin, [&](int l) { return w.B(l); }, w.ofin(),
           w.odirh(), g->inst_sigma, g->inst_rgb);
  }
  return c.rc;
}

int objnerf_mlp_generic_train_backward(const objnerf_mlp_generic_args* g, const float* d_sigma, const float* d_rgb,
                                       const float* d_inst_sigma, const float* d_inst_rgb, float* const* h_param_grads,
                                       float* d_emb_xyz, int emb_cols, float* d_obj_voxel, float* d_obj_code, float* scratch,
                                       void* stream) {
  if (!g || !g->h_params || !h_param_grads || !g->workspace || !scratch || !d_sigma || !d_rgb || !g->emb_xyz || !g->emb_dir || !g->rgb)
    return set_error(-1, "mlp_generic_train_backward: bad arguments");
  const objnerf_arch* a = &g->arch;
  if (check_arch(a)) return -1;
  if (emb_cols < 0 || emb_cols > a->in_xyz || (emb_cols > 0 && !d_emb_xyz)) return set_error(-1, "mlp_generic_train_backward: bad emb_cols");
  const bool obj = g->do_object != 0;
  if (obj && (!d_inst_sigma || !d_inst_rgb || !g->inst_rgb || (a->code_c > 0 && (!d_obj_code || !g->obj_code)) ||
              (a->obj_voxel_c > 0 && (!d_obj_voxel || !g->obj_voxel))))
    return set_error(-1, "mlp_generic_train_backward: object branch gradients missing");
  const long P = g->n_points;
  if (P <= 0) return P < 0 ? set_error(-1, "mlp_generic_train_backward: negative n_points") : 0;
  GCtx c{(hipStream_t)stream, 0};
  const GenWs w{a, P, g->workspace};
  const GenWs d{a, P, scratch};
  float* t2 = scratch + gen_ws_floats(a, P);
  float* t2i = t2 + 3 * P;
  const float* const* p = g->h_params;
  hipLaunchKernelGGL(sigmoid_bwd_generic_kernel, dim3(nblk(3 * P)), dim3(256), 0, c.s, t2, d_rgb, g->rgb, 3 * P);
  if (obj) hipLaunchKernelGGL(sigmoid_bwd_generic_kernel, dim3(nblk(3 * P)), dim3(256), 0, c.s, t2i, d_inst_rgb, g->inst_rgb, 3 * P);
  c.rc = check_launch("sigmoid_bwd");

  struct Blk { const float* x; int c; };
  // one branch: the dgrad chain through the hidden layers, every layer's weight / bias gradient, and the list of
  // (dZ_l, W_l) pairs of the layers fed by the branch's input (layer 0 and the skip layers) for the embedding gradients
  struct Seg { const float* dY; const float* W; long ldw; };
  auto branch = [&](const float* const* q, float* const* gq, int D, int W, const int32_t* skips, int nsk, const Blk* in, int nin,
                    auto act_of, auto dz_of, float* fin, float* dfin, float* dirh, float* ddirh, const float* dsig, const float* t2_,
                    Seg* segs, int& nseg) {
    int cin = 0;
    for (int i = 0; i < nin; ++i) cin += in[i].c;
    const float* const* t = q + 2 * D;          // final, dir, sigma, rgb
    float* const* gt = gq + 2 * D;
    const float* h = act_of(D - 1);
    float* dh = dz_of(D - 1);
    // heads and colour layers
    dgrad(c, t2_, 3, t[6], W / 2, P, 3, W / 2, ddirh, W / 2, 0, dirh);                       // through rgb.0 + the dir layer's LeakyReLU
    dgrad(c, ddirh, W / 2, t[2], W + a->in_dir, P, W / 2, W, dfin, W, 0);                    // -> d(final), no activation
    dgrad(c, dfin, W, t[0], W, P, W, W, dh, W, 0);                                           // -> d(h_{D-1}) ...
    dgrad(c, dsig, 1, t[4], W, P, 1, W, dh, W, 1, h);                                        // ... + d sigma * w_sigma, then leaky'
    wgrad(c, t2_, 3, dirh, W / 2, P, 3, W / 2, gt[6], W / 2, gt[7]);
    wgrad(c, ddirh, W / 2, fin, W, P, W / 2, W, gt[2], W + a->in_dir, gt[3]);
    wgrad(c, ddirh, W / 2, g->emb_dir, a->in_dir, P, W / 2, a->in_dir, gt[2] + W, W + a->in_dir);
    wgrad(c, dfin, W, h, W, P, W, W, gt[0], W, gt[1]);
    wgrad(c, dsig, 1, h, W, P, 1, W, gt[4], W, gt[5]);
    nseg = 0;
    for (int l = D - 1; l >= 0; --l) {
      const float* dZ = dz_of(l);
      const bool skip = l > 0 && has(skips, nsk, l);
      const long ldw = l == 0 ? cin : (skip ? cin + W : W);
      if (l > 0) {   // hidden block: d(h_{l-1}) = leaky'(A_{l-1}) . (dZ_l W_l[:, hidden])
        dgrad(c, dZ, W, q[2 * l] + (skip ? cin : 0), ldw, P, W, W, dz_of(l - 1), W, 0, act_of(l - 1));
        wgrad(c, dZ, W, act_of(l - 1), W, P, W, W, gq[2 * l] + (skip ? cin : 0), ldw, (skip ? nullptr : gq[2 * l + 1]));
      }
      if (l == 0 || skip) {   // input blocks: column blocks of W_l, bias with the first block
        int col = 0;
        for (int i = 0; i < nin; ++i) {
          wgrad(c, dZ, W, in[i].x, in[i].c, P, W, in[i].c, gq[2 * l] + col, ldw, i == 0 ? gq[2 * l + 1] : nullptr);
          col += in[i].c;
        }
        segs[nseg++] = Seg{dZ, q[2 * l], ldw};
      }
    }
  };
  // dX_block (P x cols) = sum over consumer layers of dZ_l W_l[:, col0 : col0 + cols] (segmented contraction, 4 per launch)
  auto input_grad = [&](const Seg* segs, int nseg, int Wd, int col0, int cols, float* dX, long ldx, int accumulate0) {
    for (int s0 = 0; s0 < nseg && !c.rc; s0 += 4) {
      const int ns = nseg - s0 < 4 ? nseg - s0 : 4;
      GemmArgs gg{segs[s0].dY, Wd, 1, segs[s0].W + col0, segs[s0].ldw, 0, dX, ldx, P, cols, Wd, (s0 > 0 || accumulate0) ? 1 : 0, EPI_NONE,
                  nullptr, 1, nullptr};
      if (ns > 1) {
        gg.nseg = ns;
        for (int i = 0; i < ns; ++i) {
          gg.segA[i] = segs[s0 + i].dY; gg.seglda[i] = Wd;
          gg.segB[i] = segs[s0 + i].W + col0; gg.segldb[i] = segs[s0 + i].ldw;
          gg.segK[i] = Wd;
        }
      }
      c.rc = gemm_launch(gg, c.s);
    }
  };
  Seg ssegs[9], osegs[9];
  int nss = 0, nos = 0;
  {
    const Blk in[1] = {{g->emb_xyz, a->in_xyz}};
    branch(p, h_param_grads, a->D, a->W, a->skips, a->n_skips, in, 1, [&](int l) { return w.A(l); }, [&](int l) { return d.A(l); },
           w.fin(), d.fin(), w.dirh(), d.dirh(), d_sigma, t2, ssegs, nss);
  }
  if (obj) {
    Blk in[3];
    int nin = 0;
    in[nin++] = {g->emb_xyz, a->in_xyz};
    if (a->obj_voxel_c > 0) in[nin++] = {g->obj_voxel, a->obj_voxel_c};
    if (a->code_c > 0) in[nin++] = {g->obj_code, a->code_c};
    const int off = 2 * (a->D + 4);
    branch(p + off, h_param_grads + off, a->inst_D, a->inst_W, a->inst_skips, a->n_inst_skips, in, nin, [&](int l) { return w.B(l); },
           [&](int l) { return d.B(l); }, w.ofin(), d.ofin(), w.odirh(), d.odirh(), d_inst_sigma, t2i, osegs, nos);
  }
  // gradients w.r.t. the inputs: the first emb_cols columns of emb_xyz (the voxel-feature part: the xyz positional encoding
  // has no consumer, depths are detached, rendering.py:307), obj_voxel, obj_code
  if (emb_cols > 0) {
    input_grad(ssegs, nss, a->W, 0, emb_cols, d_emb_xyz, emb_cols, 0);
    if (obj) input_grad(osegs, nos, a->inst_W, 0, emb_cols, d_emb_xyz, emb_cols, 1);
  }
  if (obj) {
    if (a->obj_voxel_c > 0) input_grad(osegs, nos, a->inst_W, a->in_xyz, a->obj_voxel_c, d_obj_voxel, a->obj_voxel_c, 0);
    if (a->code_c > 0) input_grad(osegs, nos, a->inst_W, a->in_xyz + a->obj_voxel_c, a->code_c, d_obj_code, a->code_c, 0);
  }
  return c.rc;
}

int objnerf_pos_encode_block_backward(const float* x, int64_t ldx, int64_t n, int C, int n_freqs, const float* freqs,
                                      const float* d_out, int64_t ldo, float* d_x, int64_t ldd, void* stream) {
  if (!x || !d_out || !d_x || C < 1 || n_freqs < 0 || ldx < C || ldd < C || ldo < (int64_t)C * (2 * n_freqs + 1))
    return set_error(-1, "pos_encode_block_backward: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(pos_encode_block_bwd_kernel, dim3(nblk(n * C)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (long)n, C, n_freqs,
                     freqs, d_out, (long)ldo, d_x, (long)ldd);
  return check_launch("pos_encode_block_backward");
}

int objnerf_voxel_features_backward(const objnerf_voxel_grid* grid, int C, const float* xyz, int64_t n, const float* d_raw,
                                    int64_t ldd, float* table_grad, void* stream) {
  if (!grid || !grid->idx_map || !xyz || !d_raw || !table_grad || C < 1 || ldd < C) return set_error(-1, "voxel_features_backward: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(voxel_features_bwd_kernel, dim3(nblk(n * C)), dim3(256), 0, (hipStream_t)stream, *grid, C, xyz, (long)n, d_raw,
                     (long)ldd, table_grad);
  return check_launch("voxel_features_backward");
}

}  // extern "C"
