// train_kernels.hip -- backward kernels of the HBM-bound stages:
// alpha-compositing backward (one wave per ray), voxel-embedding backward (fp32 atomics into the
// feature table), the per-ray reduction of object-code gradients and the materialised sample points.
#include <hip/hip_runtime.h>
#include "layout.h"
#include "device_math.h"
#include "host_api.h"

namespace objnerf {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wscan_mul(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(v, o); if (lane >= o) v *= t; }
  return v;
}
__device__ __forceinline__ float wscan_add(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(v, o); if (lane >= o) v += t; }
  return v;
}

// Backward of one composited channel set (models/rendering.py:156-176 / 187-223).
//   w_i = a_i T_i,  T_i = prod_{j<i} (1 - a_j + 1e-10),  a_i = 1 - exp(-delta_i relu(s_i))
//   L depends on  C = sum w_i c_i,  D = sum w_i z_i,  O = sum w_i     (go already holds dL/dO minus the
//   white-background term).  With g_i = gC . c_i + gD z_i + go:
//     dL/da_i = T_i g_i - (sum_{j>i} w_j g_j) / (1 - a_i + 1e-10)
//     dL/ds_i = dL/da_i * delta_i * exp(-delta_i relu(s_i)) * [s_i > 0]          (0 where the alpha was masked)
//     dL/dc_i = w_i gC
// Two forward sweeps.  The first leaves the sum of w_j g_j over chunk c (64 samples) in lane c of `vchunk`; the second
// builds  sum_{j>i} w_j g_j  = (chunks after i's) + (exclusive REVERSE scan inside i's chunk) from those -- sums of the
// terms behind sample i only.  (Taking it as V - prefix cancels behind an opaque surface, where the suffix is ~0 and is
// then divided by t ~ 1e-10.)  No per-ray storage; S <= 4096.
__device__ __forceinline__ float wscan_add_rev_excl(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_down(v, o); if (lane + o < 64) v += t; }
  const float e = __shfl_down(v, 1);
  return lane == 63 ? 0.f : e;
}
__device__ __forceinline__ void composite_bwd_ray(const float* __restrict__ z, const float* __restrict__ sigma,
                                                  const float* __restrict__ rgb, const float* __restrict__ noise,
                                                  float noise_std, float last_delta, int S, int lane, bool occl,
                                                  float occl_limit, float gC0, float gC1, float gC2, float gD, float go,
                                                  float* __restrict__ d_sigma, float* __restrict__ d_rgb) {
  float vchunk = 0.f;
  for (int sweep = 0; sweep < 2; ++sweep) {
    float carry = 1.f;                     // transmittance in front of the chunk
    for (int base = 0, c = 0; base < S; base += 64, ++c) {
      const int i = base + lane;
      const bool in = i < S;
      float zi = 0.f, alpha = 0.f, dads = 0.f, gi = 0.f;
      if (in) {
        zi = z[i];
        const float delta = i + 1 < S ? z[i + 1] - zi : last_delta;
        float s = sigma[i];
        if (noise) s = s + noise[i] * noise_std;
        const float e = expf(-delta * fmaxf(s, 0.f));
        alpha = 1.f - e;
        dads = s > 0.f ? delta * e : 0.f;
        if (occl && occl_limit < zi) { alpha = 0.f; dads = 0.f; }
        gi = gC0 * rgb[i * 3] + gC1 * rgb[i * 3 + 1] + gC2 * rgb[i * 3 + 2] + gD * zi + go;
      }
      const float t = in ? (1.f - alpha) + 1e-10f : 1.f;
      const float incl = wscan_mul(t, lane);
      float excl = __shfl_up(incl, 1);
      if (lane == 0) excl = 1.f;
      const float T = carry * excl;
      const float w = alpha * T;
      const float v = in ? w * gi : 0.f;
      if (sweep == 0) {
        const float tot = wsum(v);
        if (lane == c) vchunk = tot;
      } else {
        const float later = wsum(lane > c ? vchunk : 0.f);     // chunks behind this one
        const float suffix = later + wscan_add_rev_excl(v, lane);
        if (in) {
          const float dLda = T * gi - suffix / t;
          d_sigma[i] = dLda * dads;
          d_rgb[i * 3] = w * gC0; d_rgb[i * 3 + 1] = w * gC1; d_rgb[i * 3 + 2] = w * gC2;
        }
      }
      carry = carry * __shfl(incl, 63);
    }
  }
}

// scene depth of one ray (needed by the occlusion mask of the instance set, rendering.py:192-197)
__device__ __forceinline__ float scene_depth(const float* __restrict__ z, const float* __restrict__ sigma,
                                             const float* __restrict__ noise, float noise_std, float last_delta, int S,
                                             int lane) {
  float carry = 1.f, sd = 0.f;
  for (int base = 0; base < S; base += 64) {
    const int i = base + lane;
    const bool in = i < S;
    float zi = 0.f, alpha = 0.f;
    if (in) {
      zi = z[i];
      const float delta = i + 1 < S ? z[i + 1] - zi : last_delta;
      float s = sigma[i];
      if (noise) s = s + noise[i] * noise_std;
      alpha = 1.f - expf(-delta * fmaxf(s, 0.f));
    }
    const float t = in ? (1.f - alpha) + 1e-10f : 1.f;
    const float incl = wscan_mul(t, lane);
    float excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 1.f;
    if (in) sd += alpha * (carry * excl) * zi;
    carry = carry * __shfl(incl, 63);
  }
  return wsum(sd);
}

struct CompBwd {
  objnerf_composite_args f;
  const float *g_rgb, *g_depth, *g_opac, *g_rgb_i, *g_depth_i, *g_opac_i;
  float *d_sigma, *d_rgb, *d_isigma, *d_irgb;
};

__global__ void __launch_bounds__(256) composite_bwd_kernel(const CompBwd b) {
  const objnerf_composite_args& a = b.f;
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const int S = a.S;
  for (long ray = wave; ray < a.n_rays; ray += nwaves) {
    const float* z = a.z_vals + ray * S;
    const float* nz = a.noise ? a.noise + ray * S : nullptr;
    const float last = a.use_zero_as_last_delta ? 0.f : 1e10f;
    {
      const float g0 = b.g_rgb ? b.g_rgb[ray * 3] : 0.f, g1 = b.g_rgb ? b.g_rgb[ray * 3 + 1] : 0.f,
                  g2 = b.g_rgb ? b.g_rgb[ray * 3 + 2] : 0.f;
      const float gD = b.g_depth ? b.g_depth[ray] : 0.f;
      // rgb_map += 1 - opacity when white_back (rendering.py:178-179)
      const float go = (b.g_opac ? b.g_opac[ray] : 0.f) - (a.white_back ? g0 + g1 + g2 : 0.f);
      composite_bwd_ray(z, a.sigma + ray * S, a.rgb + ray * S * 3, nz, a.noise_std, last, S, lane, false, 0.f, g0, g1, g2,
                        gD, go, b.d_sigma + ray * S, b.d_rgb + ray * S * 3);
    }
    if (a.inst_sigma) {
      bool occl = a.occlusion != 0;
      if (occl && a.pass_through_mask && a.pass_through_mask[ray]) occl = false;
      float limit = 0.f;
      if (occl) limit = scene_depth(z, a.sigma + ray * S, nz, a.noise_std, last, S, lane) + a.frustum_bound_th;
      const float g0 = b.g_rgb_i ? b.g_rgb_i[ray * 3] : 0.f, g1 = b.g_rgb_i ? b.g_rgb_i[ray * 3 + 1] : 0.f,
                  g2 = b.g_rgb_i ? b.g_rgb_i[ray * 3 + 2] : 0.f;
      const float gD = b.g_depth_i ? b.g_depth_i[ray] : 0.f;
      const float go = (b.g_opac_i ? b.g_opac_i[ray] : 0.f) - (g0 + g1 + g2);     // always white-backed, :223
      composite_bwd_ray(z, a.inst_sigma + ray * S, a.inst_rgb + ray * S * 3, a.noise_inst ? a.noise_inst + ray * S : nullptr,
                        a.noise_std, 0.f, S, lane, occl, limit, g0, g1, g2, gD, go, b.d_isigma + ray * S,
                        b.d_irgb + ray * S * 3);
    }
  }
}

// ---- voxel embedding backward ---------------------------------------------------------------------
// A workgroup holds 64..192 consecutive sample points = consecutive depths of one to three rays, and neighbouring depths
// fall into the same voxel cell: sent straight to memory, their 8 x 24 atomics per point pile up on a few table rows
// (the fine pass, whose samples cluster at surfaces, ran 8x slower per point than the coarse pass).  The workgroup
// therefore first sums its contributions per table row in an LDS hash (row id -> 24 floats, ds_add_f32), then sends
// each row it touched to memory once; contributions that find no slot within 8 probes go to memory directly.
#ifndef OBJ_VB_SLOT_BITS
#define OBJ_VB_SLOT_BITS 8         // 256 slots: 26 KB of LDS, six workgroups per CU (512: three) -- step -0.15 ms, profiles/r04_train_ab.txt
#endif
#ifndef OBJ_VB_PROBE_LANES
#define OBJ_VB_PROBE_LANES 1        // 1: one probing lane per corner (round 5); 0: every channel lane probes (A/B build switch)
#endif
constexpr int kVbSlots = 1 << OBJ_VB_SLOT_BITS;               // power of two
constexpr int kVbStride = kVoxC + 1;        // odd stride: spreads the rows over the LDS banks
// (points per workgroup -- the aggregation window -- are chosen per launch: launch_voxel_embed_bwd)
// 32 lanes per point, lane = voxel channel (0..15 scene, 16..23 object; 24..31 idle), 8 points per pass of a
// 256-thread workgroup: a corner's features and every (frequency, sin|cos) block of the incoming gradient row are
// contiguous across the lanes, and the 24 lanes of a point add to 24 different LDS words of its row's slot.
// SAVED (round 6, the training path): the interpolated feature f of every channel is READ BACK -- the forward stored it as the
// identity block of the positional encoding (embedding_helper.py:69-74: [f, sin, cos, ...]; columns 0..15 of the emb_xyz row, 0..7 of
// the obj_voxel row: the same bits this kernel would recompute) -- instead of gathered again from the 8 corner rows of the table:
// the 8 x 96-byte random reads per point and one of the kernel's three dependent round trips (position -> index map -> table rows)
// disappear; the index-map reads remain (the scatter's destinations).
template <bool SAVED>
__global__ void __launch_bounds__(256) voxel_embed_bwd_kernel(const objnerf_voxel_grid g, const float* __restrict__ xyz,
                                                               long n, const float* __restrict__ d_scene,
                                                               const float* __restrict__ d_obj,
                                                               float* __restrict__ table_grad,
                                                               const float* __restrict__ f_scene, const float* __restrict__ f_obj,
                                                               const int ppw) {          // points per workgroup: a multiple of 8
  __shared__ int keys[kVbSlots];
  __shared__ float vals[kVbSlots * kVbStride];
  for (int i = threadIdx.x; i < kVbSlots; i += 256) keys[i] = -1;
  for (int i = threadIdx.x; i < kVbSlots * kVbStride; i += 256) vals[i] = 0.f;
  __syncthreads();
  const int sub = threadIdx.x & 31;
  const bool scn = sub < kScnVoxC;
  const float* dbase = scn ? d_scene : d_obj;                       // object gradients may be absent (scene-only training)
  const long dld = scn ? (kScnVoxPE + kXyzPE) : kObjVoxPE;
  const int C = scn ? kScnVoxC : kObjVoxC;
  const int cc = scn ? sub : sub - kScnVoxC;
  for (int it = 0; it < ppw / 8; ++it) {
    const long p = (long)blockIdx.x * ppw + it * 8 + (threadIdx.x >> 5);
    if (p >= n || sub >= kVoxC || !dbase) continue;
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
    int row[8];
    const float X = (float)g.shape[0], Y = (float)g.shape[1], Z = (float)g.shape[2];
    float f = 0.f;                                                  // this channel's interpolated feature (forward value)
    if constexpr (SAVED) f = scn ? f_scene[p * (long)(kScnVoxPE + kXyzPE) + cc] : f_obj[p * (long)kObjVoxPE + cc];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float cx = qx + (float)((k >> 2) & 1), cy = qy + (float)((k >> 1) & 1), cz = qz + (float)(k & 1);
      const bool ok = cx >= 0.f && cx < X && cy >= 0.f && cy < Y && cz >= 0.f && cz < Z;
      int r = -1;
      if (ok) {
        r = g.idx_map[((size_t)(int)cx * g.shape[1] + (int)cy) * g.shape[2] + (int)cz];
        if (r >= g.n_rows) r = -1;
      }
      row[k] = r;
      if constexpr (!SAVED) { if (r >= 0) f = f + g.table[(size_t)r * kVoxC + sub] * wt[k]; }
    }
    // d/df [f, sin(2^k f), cos(2^k f)]: 1, 2^k cos, -2^k sin
    const float* d = dbase + p * dld;
    float dF = d[cc];
    float fr = 1.f;
    for (int k = 0; k < kFreqVox; ++k) {
      const SinCos sc = psincos<true>(fr * f);
      dF += fr * (d[C * (1 + 2 * k) + cc] * sc.c - d[C * (2 + 2 * k) + cc] * sc.s);
      fr *= 2.f;
    }
#if OBJ_VB_PROBE_LANES
    // slot of each corner's row: lane k of the point's 32 probes for corner k and the point's lanes read the result (round 5;
    // until then all 24 channel lanes ran the same compare-and-swap on the same key word: 24 serialised LDS atomics per probe --
    // the kernel's 0.33 bank conflicts per LDS cycle, profiles/r04_train_pmc.md)
    int my_row = -1;
#pragma unroll
    for (int k = 0; k < 8; ++k) my_row = sub == k ? row[k] : my_row;
    int my_slot = -1;
    if (sub < 8 && my_row >= 0) {
      unsigned h = ((unsigned)my_row * 2654435761u) >> (32 - OBJ_VB_SLOT_BITS);
      for (int probe = 0; probe < 8; ++probe) {
        const int prev = atomicCAS(&keys[h], -1, my_row);
        if (prev == -1 || prev == my_row) { my_slot = (int)h; break; }
        h = (h + 1) & (kVbSlots - 1);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int slot = __shfl(my_slot, k, 32);
      if (row[k] < 0) continue;                       // invalid corners were zeroed in the forward pass
      float* t = slot >= 0 ? vals + slot * kVbStride : table_grad + (size_t)row[k] * kVoxC;   // no free slot: to memory
      atomicAdd(t + sub, dF * wt[k]);
    }
#else
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (row[k] < 0) continue;
      unsigned h = ((unsigned)row[k] * 2654435761u) >> (32 - OBJ_VB_SLOT_BITS);
      int slot = -1;
      for (int probe = 0; probe < 8; ++probe) {
        const int prev = atomicCAS(&keys[h], -1, row[k]);
        if (prev == -1 || prev == row[k]) { slot = (int)h; break; }
        h = (h + 1) & (kVbSlots - 1);
      }
      float* t = slot >= 0 ? vals + slot * kVbStride : table_grad + (size_t)row[k] * kVoxC;
      atomicAdd(t + sub, dF * wt[k]);
    }
#endif
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kVbSlots * kVoxC; i += 256) {
    const int slot = i / kVoxC, c = i - slot * kVoxC;
    const int r = keys[slot];
    if (r < 0) continue;
    const float v = vals[slot * kVbStride + c];
    if (v != 0.f) atomicAdd(table_grad + (size_t)r * kVoxC + c, v);
  }
}

__global__ void sum_over_samples_kernel(const float* __restrict__ x, long n_rays, int S, int C, float* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rays * C) return;
  const long r = idx / C;
  const int c = (int)(idx - r * C);
  float s = 0.f;
  for (int i = 0; i < S; ++i) s += x[(r * S + i) * C + c];
  out[idx] = out[idx] + s;      // accumulates: the coarse and the fine pass add into one (N, C) gradient
}

// gradient of a row gather out[i] = table[ids[i]] (nn.Embedding, models/code_library.py:20-28): one workgroup of 16 waves per
// TABLE row.  A wave takes every 16th block of 128 ids, finds the rows that picked this table row with two ballots, and adds them
// in ascending order with eight independent 256-byte row loads in flight (lane = column); the 16 waves' sums are then added in wave
// order: a FIXED summation order (bit-reproducible run to run), no atomics.  The tables are small (N_max_objs = 64 rows), the batch
// a few thousand rays.  (First version: one thread per column walking all ids through LDS with a uniform branch per id -- 2,048
// dependent LDS round trips, 0.22 ms, slower than torch's 0.115 ms; this one is bound by a handful of load latencies.)
__device__ __forceinline__ float gather_matches(unsigned long long m, long base, const float* __restrict__ d_rows, int C, int c,
                                                bool col_ok, float acc) {
  while (m) {                                    // uniform: m comes from a ballot
    long j[8];
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (m) { j[u] = base + __builtin_ctzll(m); m &= m - 1; ++cnt; } else j[u] = -1;
    }
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = (u < cnt && col_ok) ? d_rows[j[u] * C + c] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) if (u < cnt) acc += x[u];
  }
  return acc;
}
__global__ void __launch_bounds__(1024) rows_gather_bwd_kernel(const float* __restrict__ d_rows, const long* __restrict__ ids, long n,
                                                                int C, float* __restrict__ table_grad) {
  __shared__ float part[16][64];
  const long r = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + lane;
    const bool col_ok = c < C;
    float acc = 0.f;
    for (long base = (long)wave * 128; base < n; base += 16 * 128) {
      const long i0 = base + lane, i1 = base + 64 + lane;
      const bool h0 = i0 < n && ids[i0] == r, h1 = i1 < n && ids[i1] == r;
      const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
      acc = gather_matches(m0, base, d_rows, C, c, col_ok, acc);
      acc = gather_matches(m1, base + 64, d_rows, C, c, col_ok, acc);
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && col_ok) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) t += part[w][lane];
      table_grad[r * C + c] += t;
    }
    __syncthreads();
  }
}

__global__ void sample_points_kernel(const float* __restrict__ rays, const float* __restrict__ z, long n_rays, int S,
                                     float* __restrict__ xyz) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rays * S) return;
  const float* r = rays + (idx / S) * 8;
  const float zv = z[idx];
  xyz[idx * 3 + 0] = r[0] + r[3] * zv;      // separately rounded mul and add, rendering.py:279
  xyz[idx * 3 + 1] = r[1] + r[4] * zv;
  xyz[idx * 3 + 2] = r[2] + r[5] * zv;
}

}  // namespace objnerf

using namespace objnerf;
static inline unsigned blk(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

extern "C" {

int objnerf_composite_backward(const objnerf_composite_args* fwd, const float* g_rgb_map, const float* g_depth,
                               const float* g_opacity, const float* g_rgb_inst, const float* g_depth_inst,
                               const float* g_opacity_inst, float* d_sigma, float* d_rgb, float* d_inst_sigma,
                               float* d_inst_rgb, void* stream) {
  if (!fwd || !fwd->z_vals || !fwd->sigma || !fwd->rgb || !d_sigma || !d_rgb)
    return set_error(-1, "composite_backward: bad arguments");
  if (fwd->S > 4096) return set_error(-1, "composite_backward: at most 4096 samples per ray");
  if (fwd->inst_sigma && (!fwd->inst_rgb || !d_inst_sigma || !d_inst_rgb))
    return set_error(-1, "composite_backward: instance gradients missing");
  if (fwd->noise_std != 0.f && (!fwd->noise || (fwd->inst_sigma && !fwd->noise_inst)))
    return set_error(-1, "composite_backward: noise_std != 0 needs the forward's noise draws");
  if (fwd->n_rays == 0) return 0;
  CompBwd b;
  b.f = *fwd;
  if (b.f.noise_std == 0.f) { b.f.noise = nullptr; b.f.noise_inst = nullptr; }
  b.g_rgb = g_rgb_map; b.g_depth = g_depth; b.g_opac = g_opacity;
  b.g_rgb_i = g_rgb_inst; b.g_depth_i = g_depth_inst; b.g_opac_i = g_opacity_inst;
  b.d_sigma = d_sigma; b.d_rgb = d_rgb; b.d_isigma = d_inst_sigma; b.d_irgb = d_inst_rgb;
  unsigned grid = (unsigned)((fwd->n_rays + 3) / 4);
  if (grid > 256u * 32u) grid = 256u * 32u;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, b);
  return check_launch("composite_backward");
}

}  // extern "C"
namespace objnerf {
// emb_xyz / obj_voxel (optional, both or neither when d_obj_ftr is given): the forward's embedding rows of the same points -- their
// identity blocks are the interpolated features (SAVED above)
int launch_voxel_embed_bwd(const objnerf_voxel_grid* grid, const float* xyz, long n, const float* d_scene_ftr, const float* d_obj_ftr,
                           float* table_grad, const float* emb_xyz, const float* obj_voxel, hipStream_t s) {
  if (!grid || !grid->idx_map || !grid->table || !xyz || !d_scene_ftr || !table_grad)
    return set_error(-1, "voxel_embed_backward: bad arguments");
  if (n == 0) return 0;
  // (read on every call, not cached: tests/test_gpu_train.py switches it inside one process to cross-check the two forms)
  const bool saved_on = [] { const char* e = getenv("OBJNERF_SCATTER_SAVED"); return !e || atoi(e) != 0; }();
  // The aggregation window (points per workgroup).  A workgroup's time is its trips through the point loop (8 points each, a chain
  // of dependent round trips per trip), and six workgroups fit a CU (26 KB of LDS each): with the fixed 128-point window of rounds
  // 3-5 the fine pass of the reference batch launched 2,048 workgroups on 1,536 slots -- two rounds of 16 trips, the second a third
  // full -- and the coarse pass 1,024 (a third of the slots idle).  Now the window is what fills the slots ONCE: n / (6 CUs) points,
  // between a quarter and three quarters of the hash's slot count: 64 and 192 (the 256-slot hash holds the rows of about that many neighbouring points; beyond it contributions go to
  // memory one by one): 22 trips instead of 32, 11 instead of 16.  OBJNERF_VB_POINTS fixes it (tuning).
  static const int cus_n = [] {
    int dev = 0, c = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
    return c > 0 ? c : 256;
  }();
  static const int fixed_ppw = [] { const char* e = getenv("OBJNERF_VB_POINTS"); return e ? atoi(e) : 0; }();
  constexpr long kWgPerCu = (160 * 1024) / ((long)kVbSlots * (kVbStride + 1) * 4);      // resident workgroups per CU (LDS-bound): 6
  long ppw = fixed_ppw > 0 ? fixed_ppw : (n + kWgPerCu * cus_n - 1) / (kWgPerCu * cus_n);
  ppw = (ppw + 7) / 8 * 8;
  if (fixed_ppw <= 0) ppw = ppw < kVbSlots / 4 ? kVbSlots / 4 : (ppw > 3 * kVbSlots / 4 ? 3 * kVbSlots / 4 : ppw);
  if (saved_on && emb_xyz && (obj_voxel || !d_obj_ftr))
    hipLaunchKernelGGL(voxel_embed_bwd_kernel<true>, dim3(blk(n, (int)ppw)), dim3(256), 0, s, *grid, xyz, n, d_scene_ftr, d_obj_ftr,
                       table_grad, emb_xyz, obj_voxel, (int)ppw);
  else
    hipLaunchKernelGGL(voxel_embed_bwd_kernel<false>, dim3(blk(n, (int)ppw)), dim3(256), 0, s, *grid, xyz, n, d_scene_ftr, d_obj_ftr,
                       table_grad, nullptr, nullptr, (int)ppw);
  return check_launch("voxel_embed_backward");
}
}  // namespace objnerf
extern "C" {

int objnerf_voxel_embed_backward(const objnerf_voxel_grid* grid, const float* xyz, int64_t n, const float* d_scene_ftr,
                                 const float* d_obj_ftr, float* table_grad, void* stream) {
  return launch_voxel_embed_bwd(grid, xyz, (long)n, d_scene_ftr, d_obj_ftr, table_grad, nullptr, nullptr, (hipStream_t)stream);
}

int objnerf_sum_over_samples(const float* x, int64_t n_rays, int S, int C, float* out, void* stream) {
  if (!x || !out || S < 1 || C < 1) return set_error(-1, "sum_over_samples: bad arguments");
  if (n_rays == 0) return 0;
  hipLaunchKernelGGL(sum_over_samples_kernel, dim3(blk(n_rays * C, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (long)n_rays, S, C, out);
  return check_launch("sum_over_samples");
}

int objnerf_rows_gather_backward(const float* d_rows, const int64_t* ids, int64_t n, int C, int64_t n_table_rows, float* table_grad,
                                 void* stream) {
  if (!d_rows || !ids || !table_grad || C < 1 || C > 1024 || n < 0 || n_table_rows < 0)
    return set_error(-1, "rows_gather_backward: bad arguments (1 <= C <= 1024)");
  if (n == 0 || n_table_rows == 0) return 0;
  static_assert(sizeof(long) == sizeof(int64_t), "ids are int64");
  hipLaunchKernelGGL(rows_gather_bwd_kernel, dim3((unsigned)n_table_rows), dim3(1024), 0, (hipStream_t)stream, d_rows,
                     (const long*)ids, (long)n, C, table_grad);
  return check_launch("rows_gather_backward");
}

int objnerf_sample_points(const float* rays, const float* z_vals, int64_t n_rays, int S, float* xyz, void* stream) {
  if (!rays || !z_vals || !xyz || S < 1) return set_error(-1, "sample_points: bad arguments");
  if (n_rays == 0) return 0;
  hipLaunchKernelGGL(sample_points_kernel, dim3(blk(n_rays * S, 256)), dim3(256), 0, (hipStream_t)stream, rays, z_vals,
                     (long)n_rays, S, xyz);
  return check_launch("sample_points");
}

}  // extern "C"
