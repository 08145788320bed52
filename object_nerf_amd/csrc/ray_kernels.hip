// ray_kernels.hip -- the HBM-bound stages around the MLP: coarse depth sampling, alpha
// compositing (scene + instance), inverse-CDF importance sampling + merge, the multi-object
// sort/composite and the oriented-box masks, plus the stand-alone embedding kernels that back
// Embedding.forward / EmbeddingVoxel.forward.  One wave (64 lanes) owns one ray in every
// per-ray kernel: lane i holds samples i, i+64, ...; scans and reductions are wave-level
// (DPP/permute), no LDS traffic except where a per-ray table is searched (cdf, merge).
//
// Compiled with -ffp-contract=off; see device_math.h.
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "layout.h"
#include "device_math.h"
#include "composite_seg.h"
#include "host_api.h"

namespace objnerf {

// ------------------------------------------------------------------------------------------
// wave helpers
// ------------------------------------------------------------------------------------------
// float64 wave primitives (two 32-bit lane moves per value)
__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
  const long long b = __double_as_longlong(v);
  const int lo = __shfl_xor((int)(b & 0xffffffffll), m), hi = __shfl_xor((int)(b >> 32), m);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double shfl_up_f64(double v, int d) {
  const long long b = __double_as_longlong(v);
  const int lo = __shfl_up((int)(b & 0xffffffffll), d), hi = __shfl_up((int)(b >> 32), d);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double shfl_f64(double v, int src) {
  const long long b = __double_as_longlong(v);
  const int lo = __shfl((int)(b & 0xffffffffll), src), hi = __shfl((int)(b >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// (dpp<> and the 32-lane segment primitives of the compositing kernels: composite_seg.h)
// sum over the 64 lanes, returned in every lane: butterfly inside each row of 16 (quad swaps, row rotations), then the
// row totals travel row_bcast:15 -> rows 1, 3 and row_bcast:31 -> rows 2, 3; lane 63 holds the total
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp<0xB1>(0.f, v);            // quad_perm [1,0,3,2]
  v += dpp<0x4E>(0.f, v);            // quad_perm [2,3,0,1]
  v += dpp<0x124>(0.f, v);           // row_ror:4
  v += dpp<0x128>(0.f, v);           // row_ror:8   -> every lane of a row holds the row's sum
  v += dpp<0x142, 0xa>(0.f, v);      // row_bcast:15 into rows 1 and 3
  v += dpp<0x143, 0xc>(0.f, v);      // row_bcast:31 into rows 2 and 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// inclusive product scan over the 64 lanes: Kogge-Stone inside each row of 16 (row_shr 1, 2, 4, 8; lanes without a source
// multiply by 1), then the row totals as above
__device__ __forceinline__ float wave_scan_mul(float v, int /*lane*/) {
  v *= dpp<0x111>(1.f, v);
  v *= dpp<0x112>(1.f, v);
  v *= dpp<0x114>(1.f, v);
  v *= dpp<0x118>(1.f, v);
  v *= dpp<0x142, 0xa>(1.f, v);
  v *= dpp<0x143, 0xc>(1.f, v);
  return v;
}
// float64 versions (two 32-bit DPP moves per step)
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(double old, double v) {
  const long long bo = __double_as_longlong(old), bv = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp((int)(bo & 0xffffffffll), (int)(bv & 0xffffffffll), CTRL, ROW_MASK, BANK_MASK, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(bo >> 32), (int)(bv >> 32), CTRL, ROW_MASK, BANK_MASK, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_last_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_f64<0xB1>(0.0, v);
  v += dpp_f64<0x4E>(0.0, v);
  v += dpp_f64<0x124>(0.0, v);
  v += dpp_f64<0x128>(0.0, v);
  v += dpp_f64<0x142, 0xa>(0.0, v);
  v += dpp_f64<0x143, 0xc>(0.0, v);
  return wave_last_f64(v);
}
// inclusive sum scan over the 64 lanes
__device__ __forceinline__ double wave_scan_add_f64(double v) {
  v += dpp_f64<0x111>(0.0, v);
  v += dpp_f64<0x112>(0.0, v);
  v += dpp_f64<0x114>(0.0, v);
  v += dpp_f64<0x118>(0.0, v);
  v += dpp_f64<0x142, 0xa>(0.0, v);
  v += dpp_f64<0x143, 0xc>(0.0, v);
  return v;
}
// v of the lane below (lane 0 gets `fill`), and v of lane 63 in every lane
__device__ __forceinline__ float wave_shr1(float v, float fill) { return dpp<0x138>(fill, v); }
__device__ __forceinline__ float wave_last(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }

// ------------------------------------------------------------------------------------------
// K1: coarse depths (models/rendering.py:260-277)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float coarse_z(float near, float far, float t, int use_disp) {
  if (!use_disp) return near * (1.f - t) + far * t;                       // rendering.py:262
  return __fdiv_rn(1.f, __fdiv_rn(1.f, near) * (1.f - t) + __fdiv_rn(1.f, far) * t);   // :264
}

__global__ void sample_coarse_kernel(const float* __restrict__ rays, const float* __restrict__ z_steps,
                                     const float* __restrict__ perturb_rand, float perturb, int use_disp,
                                     long n_rays, int S, float* __restrict__ z_vals) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rays * S) return;
  const long ray = idx / S;
  const int s = (int)(idx - ray * S);
  const float near = rays[ray * 8 + 6], far = rays[ray * 8 + 7];
  float z = coarse_z(near, far, z_steps[s], use_disp);
  if (perturb > 0.f) {                                                   // rendering.py:268-277
    const float zl = s > 0 ? coarse_z(near, far, z_steps[s - 1], use_disp) : z;
    const float zu = s < S - 1 ? coarse_z(near, far, z_steps[s + 1], use_disp) : z;
    const float lower = s > 0 ? 0.5f * (zl + z) : z;
    const float upper = s < S - 1 ? 0.5f * (z + zu) : z;
    z = lower + (upper - lower) * (perturb * perturb_rand[idx]);
  }
  z_vals[idx] = z;
}

// unperturbed depths, four per thread (16-byte stores); same expression per element
__global__ void sample_coarse4_kernel(const float* __restrict__ rays, const float* __restrict__ z_steps, int use_disp,
                                      long n_rays, int S, float* __restrict__ z_vals) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = S >> 2;
  if (idx >= n_rays * q) return;
  // (ray, quarter-row piece) of this thread without a 64-bit division (that division was a third of the kernel's instructions:
  // 25 us = 0.44 of the store roofline in rounds 3-5): a shift when S / 4 is a power of two (every shipped config), else 32-bit
  long ray;
  if ((q & (q - 1)) == 0) ray = idx >> (31 - __builtin_clz(q));
  else if (idx < (1L << 31)) ray = (long)((unsigned)idx / (unsigned)q);
  else ray = idx / q;
  const int s = 4 * (int)(idx - ray * q);
  const float near = rays[ray * 8 + 6], far = rays[ray * 8 + 7];
  const f32x4 t = *(const f32x4*)(z_steps + s);
  f32x4 z;
#pragma unroll
  for (int j = 0; j < 4; ++j) z[j] = coarse_z(near, far, t[j], use_disp);
#ifdef OBJ_COARSE_NT
  __builtin_nontemporal_store(z, (f32x4*)(z_vals + ray * S + s));
#else
  *(f32x4*)(z_vals + ray * S + s) = z;
#endif
}

// ------------------------------------------------------------------------------------------
// Embedding.forward (embedding_helper.py:57-74)
// ------------------------------------------------------------------------------------------
// freqs == nullptr: frequency bands 2^k (logscale=True, embedding_helper.py:52-53); else the F bands read from memory
// (logscale=False: torch.linspace(1, 2^(F-1), F), embedding_helper.py:54-55) -- `freq * x` is one fp32 multiply either way
__global__ void pos_encode_kernel(const float* __restrict__ x, long n, int C, int F, const float* __restrict__ freqs,
                                  float* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * C) return;
  const long row = idx / C;
  const int c = (int)(idx - row * C);
  const float v = x[idx];
  float* o = out + row * (long)C * (2 * F + 1);
  o[c] = v;
  float f = 1.f;
  for (int k = 0; k < F; ++k) {
    const SinCos sc = psincos<true>((freqs ? freqs[k] : f) * v);
    o[C * (1 + 2 * k) + c] = sc.s;
    o[C * (2 + 2 * k) + c] = sc.c;
    f *= 2.f;
  }
}

// ------------------------------------------------------------------------------------------
// EmbeddingVoxel.forward (embedding_helper.py:325-411).  32 lanes per point, lane = one input channel of the
// positional encoding: 0..15 scene voxel features, 16..23 object voxel features, 24..26 x, y, z (27..31 idle).
// A corner's 24 features are then one 96-byte read, and every (frequency, sin|cos) block of the output row is
// written by neighbouring lanes; the per-channel arithmetic (corner order, sin/cos) is the same as a serial loop.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) voxel_embed_kernel(const objnerf_voxel_grid g, const float* __restrict__ xyz, long n,
                                                          float* __restrict__ scene_ftr, float* __restrict__ obj_ftr) {
  const int sub = threadIdx.x & 31;
  const long p = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (p >= n || sub >= kVoxC + 3) return;
  float* so = scene_ftr + p * (long)(kScnVoxPE + kXyzPE);
  float val;       // the channel this lane encodes
  float* o;        // its output block
  int C, cc, F;    // channels in the block, index inside it, number of frequencies
  if (sub >= kVoxC) {
    val = xyz[p * 3 + (sub - kVoxC)];
    o = so + kScnVoxPE; C = 3; cc = sub - kVoxC; F = kFreqXyz;
  } else {
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
      // voxel_ftr[invalid] = 0 ; (voxel_ftr * weights).sum(0)   (embedding_helper.py:351,387-389)
      const float fv = r < 0 ? 0.f : g.table[(size_t)r * kVoxC + sub];
      f = k == 0 ? fv * wt[k] : f + fv * wt[k];
    }
    val = f;
    F = kFreqVox;
    if (sub < kScnVoxC) { o = so; C = kScnVoxC; cc = sub; }
    else { o = obj_ftr + p * (long)kObjVoxPE; C = kObjVoxC; cc = sub - kScnVoxC; }
  }
  o[cc] = val;
  float fr = 1.f;
  for (int k = 0; k < F; ++k) {
    const SinCos sc = psincos<true>(fr * val);
    o[C * (1 + 2 * k) + cc] = sc.s;
    o[C * (2 + 2 * k) + cc] = sc.c;
    fr *= 2.f;
  }
}

// ------------------------------------------------------------------------------------------
// K3: alpha compositing, scene + instance (models/rendering.py:139-229).  One wave per ray.
// ------------------------------------------------------------------------------------------
struct CompositeOut { float opacity, r, g, b, depth; };

// composites one channel set over the ray (composite_seg.h: 32-sample segments, lanes 0-31 and 32-63 of a 64-sample
// chunk are two consecutive segments); when `wout` != null writes per-sample weights.
// occl_depth: when occl, alphas with (occl_depth + th) < z are zeroed (rendering.py:192-202).
__device__ __forceinline__ CompositeOut composite_ray(const float* __restrict__ z, const float* __restrict__ sigma,
                                                      const float* __restrict__ rgb, const float* __restrict__ noise,
                                                      float noise_std, float last_delta, int S, int lane,
                                                      bool occl, float occl_limit, float* __restrict__ wout) {
  RayAcc acc;
  for (int base = 0; base < S; base += 64) {
    const int i = base + lane;
    const bool in = i < S;
    float zi = 0.f, alpha = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (in) {
      zi = z[i];
      const float delta = i + 1 < S ? z[i + 1] - zi : last_delta;
      float sg_ = sigma[i];
      if (noise) sg_ = sg_ + noise[i] * noise_std;
      alpha = sample_alpha(delta, sg_);                       // rendering.py:157
      if (occl && occl_limit < zi) alpha = 0.f;
      c0 = rgb[i * 3]; c1 = rgb[i * 3 + 1]; c2 = rgb[i * 3 + 2];
    }
    SegTotals lo, hi;
    const float lw = segment_composite(alpha, in, c0, c1, c2, zi, lane, lo, hi);
    const float Tlo = acc.step(lo), Thi = acc.step(hi);       // a segment beyond S has Q = 1 and zero sums: no effect
    if (in && wout) wout[i] = (lane < 32 ? Tlo : Thi) * lw;   // rendering.py:162
  }
  CompositeOut o;
  o.opacity = acc.opacity; o.r = acc.r; o.g = acc.g; o.b = acc.b; o.depth = acc.depth;
  return o;
}

// Scene and instance set of one ray in ONE sweep (no occlusion mask, i.e. the instance alphas do not depend on the scene
// depth: eval mode, rendering.py:192): both sets' loads are in flight together, one memory round trip per chunk instead
// of two dependent ones.  Per set the arithmetic and its order are exactly composite_ray's.
__device__ __forceinline__ void composite_ray_pair(const float* __restrict__ z, const float* __restrict__ sigma0,
                                                   const float* __restrict__ rgb0, const float* __restrict__ noise0,
                                                   float last_delta0, float* __restrict__ wout0,
                                                   const float* __restrict__ sigma1, const float* __restrict__ rgb1,
                                                   const float* __restrict__ noise1, float* __restrict__ wout1,
                                                   float noise_std, int S, int lane, CompositeOut& o0, CompositeOut& o1) {
  RayAcc acc0, acc1;
  for (int base = 0; base < S; base += 64) {
    const int i = base + lane;
    const bool in = i < S;
    float zi = 0.f, al0 = 0.f, al1 = 0.f, c0[3] = {0.f, 0.f, 0.f}, c1[3] = {0.f, 0.f, 0.f};
    if (in) {
      zi = z[i];
      const bool last = i + 1 >= S;
      const float dz = last ? 0.f : z[i + 1] - zi;
      float s0 = sigma0[i], s1 = sigma1[i];
      if (noise0) s0 = s0 + noise0[i] * noise_std;
      if (noise1) s1 = s1 + noise1[i] * noise_std;
      c0[0] = rgb0[i * 3]; c0[1] = rgb0[i * 3 + 1]; c0[2] = rgb0[i * 3 + 2];
      c1[0] = rgb1[i * 3]; c1[1] = rgb1[i * 3 + 1]; c1[2] = rgb1[i * 3 + 2];
      al0 = sample_alpha(last ? last_delta0 : dz, s0);
      al1 = sample_alpha(last ? 0.f : dz, s1);                // the instance set's last delta is 0 (rendering.py:213)
    }
    SegTotals lo0, hi0, lo1, hi1;
    const float lw0 = segment_composite(al0, in, c0[0], c0[1], c0[2], zi, lane, lo0, hi0);
    const float lw1 = segment_composite(al1, in, c1[0], c1[1], c1[2], zi, lane, lo1, hi1);
    const float Tlo0 = acc0.step(lo0), Thi0 = acc0.step(hi0), Tlo1 = acc1.step(lo1), Thi1 = acc1.step(hi1);
    if (in) {
      if (wout0) wout0[i] = (lane < 32 ? Tlo0 : Thi0) * lw0;
      if (wout1) wout1[i] = (lane < 32 ? Tlo1 : Thi1) * lw1;
    }
  }
  o0.opacity = acc0.opacity; o0.r = acc0.r; o0.g = acc0.g; o0.b = acc0.b; o0.depth = acc0.depth;
  o1.opacity = acc1.opacity; o1.r = acc1.r; o1.g = acc1.g; o1.b = acc1.b; o1.depth = acc1.depth;
}

// Second half of the fused form: the MLP kernel's epilogue left per-sample local weights and per-segment records
// (objnerf_mlp_args.comp_*); one wave per ray walks the segments in ascending order -- exactly RayAcc::step as in
// composite_ray -- rescales the segment's weights in place and writes the maps.
__global__ void __launch_bounds__(256) composite_finish_kernel(const float* __restrict__ rec, long n_rays, int S, int has_inst,
                                                               int white_back, float* __restrict__ weights,
                                                               float* __restrict__ opacity, float* __restrict__ rgb_map,
                                                               float* __restrict__ depth, float* __restrict__ rgb_inst,
                                                               float* __restrict__ depth_inst, float* __restrict__ opacity_inst,
                                                               int inst_weights) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const int nseg = S >> 5;
  for (long ray = wave; ray < n_rays; ray += nwaves) {
    RayAcc s, q;
    float* w = weights + ray * S;
    // Up to 256 samples: the ray's local weights are fetched up front, all sweeps at once, before the (dependent) walk over
    // the segment records -- one wave per ray with a single load in flight left the kernel bound by latency x concurrency
    // (1.76 TB/s in round 3), not by bandwidth.  Same arithmetic, same order: bit-equal results.
    const bool pre = S <= 256;
    float wv[4] = {0.f, 0.f, 0.f, 0.f}, tw[4] = {0.f, 0.f, 0.f, 0.f};
    if (pre) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int i = jj * 64 + lane;
        if (i < S) wv[jj] = w[i];
      }
    }
    // lane l rescales sample l of segments l >> 5, l >> 5 + 2, ...: two segments per sweep of the wave
    for (int j = 0; j < nseg; j += 2) {
      const float* r0 = rec + (ray * nseg + j) * kSegRecFloats;
      const bool two = j + 1 < nseg;
      const float* r1 = two ? r0 + kSegRecFloats : r0;
      const SegTotals a0{r0[0], r0[1], r0[2], r0[3], r0[4], r0[5]};
      const SegTotals a1{two ? r1[0] : 1.f, two ? r1[1] : 0.f, two ? r1[2] : 0.f, two ? r1[3] : 0.f, two ? r1[4] : 0.f, two ? r1[5] : 0.f};
      float Tlo = s.step(a0), Thi = s.step(a1);
      if (has_inst) {
        const SegTotals b0{r0[8], r0[9], r0[10], r0[11], r0[12], r0[13]};
        const SegTotals b1{two ? r1[8] : 1.f, two ? r1[9] : 0.f, two ? r1[10] : 0.f, two ? r1[11] : 0.f, two ? r1[12] : 0.f, two ? r1[13] : 0.f};
        const float Ql = q.step(b0), Qh = q.step(b1);
        if (inst_weights) { Tlo = Ql; Thi = Qh; }
      }
      const int i = j * 32 + lane;
      if (pre) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          if (jj == (j >> 1)) tw[jj] = lane < 32 ? Tlo : Thi;
      } else if (i < S) {
        w[i] = (lane < 32 ? Tlo : Thi) * w[i];
      }
    }
    if (pre) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int i = jj * 64 + lane;
        if (i < S) w[i] = tw[jj] * wv[jj];
      }
    }
    if (lane == 0) {
      opacity[ray] = s.opacity;
      depth[ray] = s.depth;
      rgb_map[ray * 3 + 0] = white_back ? s.r + 1.f - s.opacity : s.r;
      rgb_map[ray * 3 + 1] = white_back ? s.g + 1.f - s.opacity : s.g;
      rgb_map[ray * 3 + 2] = white_back ? s.b + 1.f - s.opacity : s.b;
      if (has_inst) {
        opacity_inst[ray] = q.opacity;
        depth_inst[ray] = q.depth;
        rgb_inst[ray * 3 + 0] = q.r + 1.f - q.opacity;      // always white-backed, rendering.py:223
        rgb_inst[ray * 3 + 1] = q.g + 1.f - q.opacity;
        rgb_inst[ray * 3 + 2] = q.b + 1.f - q.opacity;
      }
    }
  }
}

// The same, laid out for memory parallelism (round 4): a workgroup takes 64 rays; every thread first puts its share of the
// 64 x S local weights in flight (16-byte loads, all independent), threads 0..63 meanwhile walk one ray's records each (the
// RayAcc chain is per ray and sequential anyway) and leave the segments' incoming transmittances in LDS; after one barrier
// the weights are rescaled and stored.  The one-wave-per-ray form above is a chain of three dependent memory round trips per
// wave (records -> weights -> store): 1.75 TB/s however the loads were ordered.  Same arithmetic in the same order: bit-equal.
constexpr int kFinishRays = 64;
__global__ void __launch_bounds__(256) composite_finish_block_kernel(const float* __restrict__ rec, long n_rays, int S, int has_inst,
                                                                     int white_back, float* __restrict__ weights,
                                                                     float* __restrict__ opacity, float* __restrict__ rgb_map,
                                                                     float* __restrict__ depth, float* __restrict__ rgb_inst,
                                                                     float* __restrict__ depth_inst, float* __restrict__ opacity_inst,
                                                                     int inst_weights) {
  extern __shared__ __attribute__((aligned(16))) float tm[];       // [ray in block][segment]: incoming transmittance
  const int tid = threadIdx.x;
  const int nseg = S >> 5;
  const long ray0 = (long)blockIdx.x * kFinishRays;
  const int nr = (int)(n_rays - ray0 < kFinishRays ? n_rays - ray0 : kFinishRays);
  f32x4* wv = (f32x4*)(weights + ray0 * S);                        // S % 32 == 0: 128-byte aligned
  const int nvec = nr * (S >> 2);
  constexpr int PRE = 16;                                          // 16-byte pieces per thread in flight (covers S <= 256)
  f32x4 pre[PRE];
#pragma unroll
  for (int k = 0; k < PRE; ++k) {
    const int v = tid + 256 * k;
    if (v < nvec) pre[k] = wv[v];
  }
  if (tid < nr) {
    const long ray = ray0 + tid;
    RayAcc s, q;
    const float* r = rec + ray * nseg * kSegRecFloats;
    for (int j = 0; j < nseg; ++j, r += kSegRecFloats) {
      const f32x4 a = *(const f32x4*)r, b = *(const f32x4*)(r + 4);
      float T = s.step(SegTotals{a[0], a[1], a[2], a[3], b[0], b[1]});
      if (has_inst) {
        const f32x4 c = *(const f32x4*)(r + 8), d = *(const f32x4*)(r + 12);
        const float Q = q.step(SegTotals{c[0], c[1], c[2], c[3], d[0], d[1]});
        if (inst_weights) T = Q;
      }
      tm[tid * nseg + j] = T;
    }
    opacity[ray] = s.opacity;
    depth[ray] = s.depth;
    rgb_map[ray * 3 + 0] = white_back ? s.r + 1.f - s.opacity : s.r;
    rgb_map[ray * 3 + 1] = white_back ? s.g + 1.f - s.opacity : s.g;
    rgb_map[ray * 3 + 2] = white_back ? s.b + 1.f - s.opacity : s.b;
    if (has_inst) {
      opacity_inst[ray] = q.opacity;
      depth_inst[ray] = q.depth;
      rgb_inst[ray * 3 + 0] = q.r + 1.f - q.opacity;      // always white-backed, rendering.py:223
      rgb_inst[ray * 3 + 1] = q.g + 1.f - q.opacity;
      rgb_inst[ray * 3 + 2] = q.b + 1.f - q.opacity;
    }
  }
  __syncthreads();
  const int vps = S >> 2;                                          // 16-byte pieces per ray
#pragma unroll
  for (int k = 0; k < PRE; ++k) {
    const int v = tid + 256 * k;
    if (v < nvec) {
      const int rl = v / vps, seg = (v - rl * vps) >> 3;           // 8 pieces per 32-sample segment
      const float t = tm[rl * nseg + seg];
      f32x4 w = pre[k];
      w[0] = t * w[0]; w[1] = t * w[1]; w[2] = t * w[2]; w[3] = t * w[3];
      wv[v] = w;
    }
  }
  for (int v = tid + 256 * PRE; v < nvec; v += 256) {              // S > 256: the rest without the early fetch
    const int rl = v / vps, seg = (v - rl * vps) >> 3;
    const float t = tm[rl * nseg + seg];
    f32x4 w = wv[v];
    w[0] = t * w[0]; w[1] = t * w[1]; w[2] = t * w[2]; w[3] = t * w[3];
    wv[v] = w;
  }
}

__global__ void __launch_bounds__(256) composite_kernel(const objnerf_composite_args a) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const int S = a.S;
  for (long ray = wave; ray < a.n_rays; ray += nwaves) {
    const float* z = a.z_vals + ray * S;
    const bool inst = a.inst_sigma != nullptr;
    // scene: last delta 1e10 unless use_zero_as_last_delta (rendering.py:143-153)
    float* wscene = (inst && a.rays_in_bbox) ? nullptr : a.weights + ray * S;
    if (inst && !a.occlusion) {          // kernel-uniform: both sets in one sweep
      CompositeOut s, q;
      composite_ray_pair(z, a.sigma + ray * S, a.rgb + ray * S * 3, a.noise ? a.noise + ray * S : nullptr,
                         a.use_zero_as_last_delta ? 0.f : 1e10f, wscene, a.inst_sigma + ray * S, a.inst_rgb + ray * S * 3,
                         a.noise_inst ? a.noise_inst + ray * S : nullptr, a.rays_in_bbox ? a.weights + ray * S : nullptr,
                         a.noise_std, S, lane, s, q);
      if (lane == 0) {
        a.opacity[ray] = s.opacity;
        a.depth[ray] = s.depth;
        a.rgb_map[ray * 3 + 0] = a.white_back ? s.r + 1.f - s.opacity : s.r;
        a.rgb_map[ray * 3 + 1] = a.white_back ? s.g + 1.f - s.opacity : s.g;
        a.rgb_map[ray * 3 + 2] = a.white_back ? s.b + 1.f - s.opacity : s.b;
        a.opacity_inst[ray] = q.opacity;
        a.depth_inst[ray] = q.depth;
        a.rgb_inst[ray * 3 + 0] = q.r + 1.f - q.opacity;
        a.rgb_inst[ray * 3 + 1] = q.g + 1.f - q.opacity;
        a.rgb_inst[ray * 3 + 2] = q.b + 1.f - q.opacity;
      }
      continue;
    }
    const CompositeOut s = composite_ray(z, a.sigma + ray * S, a.rgb + ray * S * 3,
                                         a.noise ? a.noise + ray * S : nullptr, a.noise_std,
                                         a.use_zero_as_last_delta ? 0.f : 1e10f, S, lane, false, 0.f, wscene);
    if (lane == 0) {
      a.opacity[ray] = s.opacity;
      a.depth[ray] = s.depth;
      // rgb_map + 1 - weights_sum when white_back, rendering.py:178-179
      a.rgb_map[ray * 3 + 0] = a.white_back ? s.r + 1.f - s.opacity : s.r;
      a.rgb_map[ray * 3 + 1] = a.white_back ? s.g + 1.f - s.opacity : s.g;
      a.rgb_map[ray * 3 + 2] = a.white_back ? s.b + 1.f - s.opacity : s.b;
    }
    if (inst) {
      bool occl = a.occlusion != 0;
      if (occl && a.pass_through_mask && a.pass_through_mask[ray]) occl = false;   // rendering.py:198-200
      float* winst = a.rays_in_bbox ? a.weights + ray * S : nullptr;              // rendering.py:228-229
      const CompositeOut q = composite_ray(z, a.inst_sigma + ray * S, a.inst_rgb + ray * S * 3,
                                           a.noise_inst ? a.noise_inst + ray * S : nullptr, a.noise_std,
                                           0.f, S, lane, occl, s.depth + a.frustum_bound_th, winst);
      if (lane == 0) {
        a.opacity_inst[ray] = q.opacity;
        a.depth_inst[ray] = q.depth;
        a.rgb_inst[ray * 3 + 0] = q.r + 1.f - q.opacity;      // always white-backed, rendering.py:223
        a.rgb_inst[ray * 3 + 1] = q.g + 1.f - q.opacity;
        a.rgb_inst[ray * 3 + 2] = q.b + 1.f - q.opacity;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// K4: sample_pdf (rendering.py:11-61) and the merge sort(cat([z, z_])) (rendering.py:313)
// One wave per ray, four rays per workgroup; each wave owns a slice of the dynamic LDS (bins | cdf | merge buffer |
// sorted new samples) and never synchronises with the other three (LDS operations of one wave execute in order).
// ------------------------------------------------------------------------------------------
constexpr int kMaxBins = 1024;      // supports N_samples up to 1025
constexpr int kMaxMerge = 2048;     // S + I

__device__ __forceinline__ bool wave_all(bool p) { return __ballot(p) == ~0ull; }

// builds cdf[0..nb) for weights w[0..nb-1) and draws `I` samples into out (lane-strided)
__device__ __forceinline__ void sample_pdf_ray(const float* bins_lds, float* cdf_lds, const float* __restrict__ wts,
                                               int nb, const float* __restrict__ u, int I, float eps,
                                               float* out_lds, float* __restrict__ out_glb, int lane) {
  const int nw = nb - 1;
  // weights + eps, row sum (rendering.py:30-31).  The reference's CPU kernels (ATen) sum fp32 rows with a cascade that
  // is within an ulp of the exact sum; here the row is summed in float64 and rounded once.
  double part = 0.0;
  for (int i = lane; i < nw; i += 64) part += (double)(wts[i] + eps);
  const float tot = (float)wave_sum_f64(part);
  // cdf = cat([0, cumsum(pdf)]) (rendering.py:32-33).  ATen's CPU cumsum keeps its running sum in the accumulate type of
  // fp32, which is float64, and rounds every prefix to fp32 on store.  The same here: a wave-parallel inclusive scan in
  // float64 (6 DPP steps per 64 elements + a float64 carry), each prefix rounded to fp32.  A float64 prefix of <= 1023
  // fp32 terms is exact to ~1e-16 relative in any association order, so the fp32-rounded prefixes equal the sequential
  // ones except when a float64 sum lies within that distance of an fp32 rounding boundary (probability ~1e-9 per
  // element).  Why it matters: u = 1 (the last deterministic sample) lands in the last bin or one bin earlier depending
  // on whether cdf[-1] is <= 1 or > 1 -- a whole-bin discontinuity decided by the last ulp of this sum; an
  // fp32 running sum drifts several ulps from 1 and flips that decision on ~40 % of opaque rays.
  double carry = 0.0;
  if (lane == 0) cdf_lds[0] = 0.f;
  for (int base = 0; base < nw; base += 64) {
    const int idx = base + lane;
    double v = idx < nw ? (double)__fdiv_rn(wts[idx] + eps, tot) : 0.0;
    v = wave_scan_add_f64(v) + carry;
    if (idx < nw) cdf_lds[idx + 1] = (float)v;
    carry = wave_last_f64(v);
  }
  __builtin_amdgcn_wave_barrier();
  for (int j = lane; j < I; j += 64) {
    const float uj = u[j];
    // searchsorted(cdf, u, right=True): first index with cdf[idx] > u  (rendering.py:43)
    int lo = 0, hi = nb;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf_lds[mid] <= uj) lo = mid + 1; else hi = mid;
    }
    const int below = lo - 1 < 0 ? 0 : lo - 1;       // clamp_min(inds-1, 0)
    const int above = lo > nw ? nw : lo;             // clamp_max(inds, N_samples_)
    const float c0 = cdf_lds[below], c1 = cdf_lds[above];
    const float b0 = bins_lds[below], b1 = bins_lds[above];
    float denom = c1 - c0;
    if (denom < eps) denom = 1.f;                    // rendering.py:53-54
    const float smp = b0 + __fdiv_rn(uj - c0, denom) * (b1 - b0);   // rendering.py:58-60
    if (out_lds) out_lds[j] = smp;
    if (out_glb) out_glb[j] = smp;
  }
  __builtin_amdgcn_wave_barrier();
}

__global__ void __launch_bounds__(256) sample_pdf_kernel(const float* __restrict__ bins, const float* __restrict__ weights,
                                                         const float* __restrict__ u, long u_stride, long n_rays,
                                                         int nb, int I, float eps, float* __restrict__ samples) {
  extern __shared__ float pdf_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
  float* bins_lds = pdf_lds + wave * 2 * nb;
  float* cdf_lds = bins_lds + nb;
  for (long ray = (long)blockIdx.x * nwb + wave; ray < n_rays; ray += (long)gridDim.x * nwb) {
    for (int i = lane; i < nb; i += 64) bins_lds[i] = bins[ray * nb + i];
    __builtin_amdgcn_wave_barrier();
    sample_pdf_ray(bins_lds, cdf_lds, weights + ray * (nb - 1), nb, u + ray * u_stride, I, eps,
                   nullptr, samples + ray * I, lane);
  }
}

// number of elements of the ascending array a[0..n) that are < v (strict) or <= v
__device__ __forceinline__ int count_below(const float* a, int n, float v, bool or_equal) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const bool take = or_equal ? a[mid] <= v : a[mid] < v;
    if (take) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// fine depths: z_mid bins from the coarse depths, weights[:,1:-1], then ascending merge
__global__ void __launch_bounds__(256) sample_pdf_merge_kernel(const float* __restrict__ z_coarse,
                                                               const float* __restrict__ weights,
                                                               const float* __restrict__ u, long u_stride,
                                                               long n_rays, int S, int I, float eps,
                                                               float* __restrict__ z_samples, float* __restrict__ z_fine,
                                                               const float* __restrict__ clip) {
  extern __shared__ float pdf_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
  const int nb = S - 1;
  const int M = S + I;
  float* bins_lds = pdf_lds + wave * (2 * nb + M + I);
  float* cdf_lds = bins_lds + nb;
  float* all_lds = cdf_lds + nb;         // [0,S) coarse z, [S,S+I) new samples
  float* sorted_new = all_lds + M;       // the new samples in ascending (stable) order, when they are not already
  for (long ray = (long)blockIdx.x * nwb + wave; ray < n_rays; ray += (long)gridDim.x * nwb) {
    const float* z = z_coarse + ray * S;
    bool coarse_sorted = true;
    for (int i = lane; i < S; i += 64) {
      const float zi = z[i];
      all_lds[i] = zi;
      // z_vals_mid = 0.5 * (z[:-1] + z[1:])   (rendering.py:302-304)
      if (i < nb) {
        const float zn = z[i + 1];
        bins_lds[i] = 0.5f * (zi + zn);
        coarse_sorted = coarse_sorted && zn >= zi;
      }
    }
    coarse_sorted = wave_all(coarse_sorted);
    __builtin_amdgcn_wave_barrier();
    sample_pdf_ray(bins_lds, cdf_lds, weights + ray * S + 1, nb, u + ray * u_stride, I, eps,
                   all_lds + S, z_samples ? z_samples + ray * I : nullptr, lane);
    float* out = z_fine + ray * M;
    // optional per-ray (lo, hi): merged depths strictly inside the interval are moved to hi (the 10-column ray sets of
    // render_rays_multi, multi_rendering.py:277-285); the row stays ascending (the moved run is contiguous and <= hi)
    const bool do_clip = clip != nullptr;
    const float clo = do_clip ? clip[ray * 2] : 0.f, chi = do_clip ? clip[ray * 2 + 1] : 0.f;
    auto put = [&](int pos, float v) { out[pos] = (do_clip && v > clo && v < chi) ? chi : v; };
    // z_fine = torch.sort(torch.cat([z, z_]))[0], ties in concatenation order (stable)
    if (!coarse_sorted) {
      // general case (far < near): rank of every element among all M
      for (int i = lane; i < M; i += 64) {
        const float v = all_lds[i];
        int rank = 0;
        for (int j = 0; j < M; ++j) {
          const float o = all_lds[j];
          rank += (o < v || (o == v && j < i)) ? 1 : 0;
        }
        put(rank, v);
      }
    } else {
      // Both runs ascending -> two binary searches per element.  The new samples are ascending whenever u is
      // (deterministic sampling: the inverse cdf is monotone); random u: order them first by rank among themselves.
      const float* nw_ = all_lds + S;
      bool new_sorted = true;
      for (int j = lane; j < I; j += 64) new_sorted = new_sorted && (j == 0 || nw_[j] >= nw_[j - 1]);
      new_sorted = wave_all(new_sorted);
      if (!new_sorted) {
        for (int j = lane; j < I; j += 64) {
          const float v = nw_[j];
          int rank = 0;
          for (int k = 0; k < I; ++k) {
            const float o = nw_[k];
            rank += (o < v || (o == v && k < j)) ? 1 : 0;
          }
          sorted_new[rank] = v;
        }
        __builtin_amdgcn_wave_barrier();
        nw_ = sorted_new;
      }
      // a coarse depth goes behind the new samples strictly below it; a new sample behind the coarse depths <= it
      for (int i = lane; i < S; i += 64) put(i + count_below(nw_, I, all_lds[i], false), all_lds[i]);
      for (int r = lane; r < I; r += 64) put(r + count_below(all_lds, S, nw_[r], true), nw_[r]);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// The shipped sample counts -- S = 64 coarse depths, I = 64 or 128 importance samples -- on a specialised form of the kernel
// above (round 6; the generic kernel ran 206.7 us per 640x480 pass = 0.19 of its byte roofline, bound by INSTRUCTION ISSUE:
// ~410 wave instructions per ray, two thirds of them the exec-mask loops of three divergent binary searches).  Same arithmetic,
// same order, bit-equal results (tests/test_gpu_stages.py, test_gpu_render.py); what changes:
//   * one element per lane and fixed trip counts: no strided loops, the searches are branch-free (6-8 probe / compare / select
//     steps each), the neighbour depth comes through a DPP shift instead of a second load;
//   * persistent waves: 8,192 waves walk the rays with the NEXT ray's depths and weights already in flight, the deterministic
//     u table is loaded once per wave;
//   * the unsorted cases (far < near, random u) keep the general code of the kernel above.
// count of the elements of the ascending a[0..N) that are < v (or <= v): branch-free, N a power of two
template <int N, bool OR_EQUAL>
__device__ __forceinline__ int count_below_pow2(const float* a, float v) {
  const float last = a[N - 1];
  int pos = 0;
#pragma unroll
  for (int step = N / 2; step >= 1; step >>= 1) {
    const float c = a[pos + step - 1];
    pos = (OR_EQUAL ? c <= v : c < v) ? pos + step : pos;
  }
  return (OR_EQUAL ? last <= v : last < v) ? N : pos;
}
__device__ __forceinline__ float wave_shl1(float v, float fill) { return dpp<0x130>(fill, v); }     // v of the lane above

// the unsorted cases of the merge (far < near: coarse depths not ascending; random u: new samples not ascending), as in the
// generic kernel; out of line so that the common path keeps its registers
__device__ __noinline__ void merge_unsorted(const float* all_lds, float* sorted_new, int S, int I, bool coarse_sorted, float* out,
                                            bool do_clip, float clo, float chi, int lane) {
  const int M = S + I;
  auto put = [&](int pos, float val) { out[pos] = (do_clip && val > clo && val < chi) ? chi : val; };
  if (!coarse_sorted) {
    // general case: rank of every element among all M
#pragma unroll 1
    for (int i = lane; i < M; i += 64) {
      const float val = all_lds[i];
      int rank = 0;
#pragma unroll 1
      for (int j = 0; j < M; ++j) {
        const float o = all_lds[j];
        rank += (o < val || (o == val && j < i)) ? 1 : 0;
      }
      put(rank, val);
    }
    return;
  }
  // random u: order the new samples first by their rank among themselves, then the two-run merge
  const float* nw_ = all_lds + S;
#pragma unroll 1
  for (int j = lane; j < I; j += 64) {
    const float val = nw_[j];
    int rank = 0;
#pragma unroll 1
    for (int k = 0; k < I; ++k) {
      const float o = nw_[k];
      rank += (o < val || (o == val && k < j)) ? 1 : 0;
    }
    sorted_new[rank] = val;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll 1
  for (int i = lane; i < S; i += 64) put(i + count_below(sorted_new, I, all_lds[i], false), all_lds[i]);
#pragma unroll 1
  for (int r = lane; r < I; r += 64) put(r + count_below(all_lds, S, sorted_new[r], true), sorted_new[r]);
}

template <int NI>
__global__ void __launch_bounds__(256) sample_pdf_merge64_kernel(const float* __restrict__ z_coarse, const float* __restrict__ weights,
                                                                 const float* __restrict__ u, long u_stride, long n_rays, float eps,
                                                                 float* __restrict__ z_samples, float* __restrict__ z_fine,
                                                                 const float* __restrict__ clip) {
  constexpr int S = 64, nb = 63, nw = 62, I = 64 * NI, M = S + I;
  __shared__ float lds[4][192 + 2 * I];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* bins_lds = lds[wave];
  float* cdf_lds = bins_lds + 64;
  float* all_lds = cdf_lds + 64;        // [0, S) coarse depths, [S, S + I) new samples (contiguous: the general merge indexes both)
  float* sorted_new = all_lds + M;
  const long stride = (long)gridDim.x * 4;
  long ray = (long)blockIdx.x * 4 + wave;
  if (ray >= n_rays) return;
  float zi = z_coarse[ray * S + lane];
  float wi = lane < nw ? weights[ray * S + 1 + lane] : 0.f;
  float uj[NI];
  if (u_stride == 0) {
#pragma unroll
    for (int k = 0; k < NI; ++k) uj[k] = u[lane + 64 * k];
  }
  for (; ray < n_rays; ray += stride) {
    const long nxt = ray + stride;
    float zi_n = 0.f, wi_n = 0.f;
    if (nxt < n_rays) {                 // (uniform) the next ray's row, in flight while this one is worked on
      zi_n = z_coarse[nxt * S + lane];
      wi_n = lane < nw ? weights[nxt * S + 1 + lane] : 0.f;
    }
    if (u_stride != 0) {
#pragma unroll
      for (int k = 0; k < NI; ++k) uj[k] = u[ray * u_stride + lane + 64 * k];
    }
    // z_vals_mid = 0.5 * (z[:-1] + z[1:])   (rendering.py:302-304)
    all_lds[lane] = zi;
    const float zn = wave_shl1(zi, zi);
    const bool coarse_sorted = wave_all(lane < nb ? zn >= zi : true);
    if (lane < nb) bins_lds[lane] = 0.5f * (zi + zn);
    // pdf row sum and cdf exactly as sample_pdf_ray (float64 wave sum / scan, every prefix rounded to fp32)
    const float we = wi + eps;
    const float tot = (float)wave_sum_f64(lane < nw ? (double)we : 0.0);
    double v = lane < nw ? (double)__fdiv_rn(we, tot) : 0.0;
    v = wave_scan_add_f64(v) + 0.0;
    if (lane == 0) cdf_lds[0] = 0.f;
    if (lane < nw) cdf_lds[lane + 1] = (float)v;
    __builtin_amdgcn_wave_barrier();
    float smp[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      // searchsorted(cdf, u, right=True) over the nb = 63 entries: the number of entries <= u  (rendering.py:43)
      int lo = 0;
#pragma unroll
      for (int step = 32; step >= 1; step >>= 1) lo = cdf_lds[lo + step - 1] <= uj[k] ? lo + step : lo;
      const int below = lo - 1 < 0 ? 0 : lo - 1;       // clamp_min(inds-1, 0)
      const int above = lo > nw ? nw : lo;             // clamp_max(inds, N_samples_)
      const float c0 = cdf_lds[below], c1 = cdf_lds[above];
      const float b0 = bins_lds[below], b1 = bins_lds[above];
      float denom = c1 - c0;
      if (denom < eps) denom = 1.f;                    // rendering.py:53-54
      smp[k] = b0 + __fdiv_rn(uj[k] - c0, denom) * (b1 - b0);   // rendering.py:58-60
      all_lds[S + lane + 64 * k] = smp[k];
      if (z_samples) z_samples[ray * I + lane + 64 * k] = smp[k];
    }
    __builtin_amdgcn_wave_barrier();
    float* out = z_fine + ray * M;
    const bool do_clip = clip != nullptr;
    const float clo = do_clip ? clip[ray * 2] : 0.f, chi = do_clip ? clip[ray * 2 + 1] : 0.f;
    auto put = [&](int pos, float val) { out[pos] = (do_clip && val > clo && val < chi) ? chi : val; };
    const float* nw_ = all_lds + S;
    bool new_sorted = true;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int j = lane + 64 * k;
      new_sorted = new_sorted && (j == 0 || smp[k] >= nw_[j == 0 ? 0 : j - 1]);
    }
    new_sorted = wave_all(new_sorted);
    if (coarse_sorted && new_sorted) {
      // z_fine = torch.sort(torch.cat([z, z_]))[0], ties in concatenation order (stable): a coarse depth goes behind the new
      // samples strictly below it, a new sample behind the coarse depths <= it
      put(lane + count_below_pow2<I, false>(nw_, zi), zi);
#pragma unroll
      for (int k = 0; k < NI; ++k) put(lane + 64 * k + count_below_pow2<S, true>(all_lds, smp[k]), smp[k]);
    } else {
      merge_unsorted(all_lds, sorted_new, S, I, coarse_sorted, out, do_clip, clo, chi, lane);     // (rare: kept out of line)
    }
    __builtin_amdgcn_wave_barrier();
    zi = zi_n; wi = wi_n;
  }
}

// ------------------------------------------------------------------------------------------
// multi-object path (render_tools/multi_rendering.py)
// ------------------------------------------------------------------------------------------
// oriented boxes: BOX_DOUBLES doubles per box:
//   [0] scale_factor, [1..9] R_avg, [10..12] t_avg, [13..21] R_box, [22..24] t_box, [25..27] min, [28..30] max
// x_b = R_box (R_avg (xyz*scale) + t_avg) + t_box in float64, rounded to fp32, compared against the
// fp32-rounded bounds: utils/bbox_utils.py:119-130 (numpy float64) then 169-186 (torch fp32).
__device__ __forceinline__ bool in_any_box(float x, float y, float z, const double* __restrict__ boxes, int n_boxes) {
  bool in = false;
  for (int b = 0; b < n_boxes; ++b) {
    const double* B = boxes + (size_t)b * OBJNERF_BOX_DOUBLES;
    const double sx = (double)x * B[0], sy = (double)y * B[0], sz = (double)z * B[0];
    const double ax = B[1] * sx + B[2] * sy + B[3] * sz + B[10];
    const double ay = B[4] * sx + B[5] * sy + B[6] * sz + B[11];
    const double az = B[7] * sx + B[8] * sy + B[9] * sz + B[12];
    const float bx = (float)(B[13] * ax + B[14] * ay + B[15] * az + B[22]);
    const float by = (float)(B[16] * ax + B[17] * ay + B[18] * az + B[23]);
    const float bz = (float)(B[19] * ax + B[20] * ay + B[21] * az + B[24]);
    const bool ib = bx >= (float)B[25] && bx <= (float)B[28] && by >= (float)B[26] && by <= (float)B[29] &&
                    bz >= (float)B[27] && bz <= (float)B[30];
    in = in || ib;
  }
  return in;
}

__global__ void mask_sigma_kernel(float* __restrict__ sigma, float* __restrict__ rgb, const float* __restrict__ rays,
                                  const float* __restrict__ z_vals, long n_rays, int S,
                                  const double* __restrict__ boxes, int n_boxes) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rays * S) return;
  const long ray = idx / S;
  bool kill = z_vals[ray * S + S - 1] == 0.f;          // zero_mask, multi_rendering.py:40,83,92
  if (kill && rgb) {
    // the ray was not evaluated at all (objnerf_compact_rays): its colours are never weighted (alpha = 0 exactly) but
    // must be finite for 0 * rgb
    rgb[idx * 3] = 0.f; rgb[idx * 3 + 1] = 0.f; rgb[idx * 3 + 2] = 0.f;
  }
  if (!kill && n_boxes > 0) {
    const float* r = rays + ray * 8;
    const float zv = z_vals[idx];
    kill = in_any_box(r[0] + r[3] * zv, r[1] + r[4] * zv, r[2] + r[5] * zv, boxes, n_boxes);   // :239-241
  }
  if (kill) sigma[idx] = -1e5f;
}

// Ray culling on the device (no host round trip): the rays whose last depth is non-zero, in ascending order.
// Two launches: per-block counts, then every block sums the counts in front of it and writes its rays.
constexpr int kCompactBlock = 1024;
__global__ void __launch_bounds__(kCompactBlock) ray_count_kernel(const float* __restrict__ z_vals, long n_rays, int S,
                                                                  int* __restrict__ block_counts) {
  __shared__ int wave_cnt[kCompactBlock / 64];
  const long ray = (long)blockIdx.x * kCompactBlock + threadIdx.x;
  const bool on = ray < n_rays && !(z_vals[ray * S + S - 1] == 0.f);
  const unsigned long long m = __ballot(on);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < kCompactBlock / 64; ++w) t += wave_cnt[w];
    block_counts[blockIdx.x] = t;
  }
}
__global__ void __launch_bounds__(kCompactBlock) ray_compact_kernel(const float* __restrict__ z_vals, long n_rays, int S,
                                                                    const int* __restrict__ block_counts,
                                                                    int* __restrict__ ray_index, int* __restrict__ n_active) {
  __shared__ int wave_cnt[kCompactBlock / 64];
  __shared__ int part[kCompactBlock / 64];
  __shared__ int base_sh;
  // rays in front of this block, and (block 0) the grand total
  int mine = 0, all = 0;
  for (int b = threadIdx.x; b < (int)gridDim.x; b += kCompactBlock) {
    const int c = block_counts[b];
    all += c;
    mine += b < (int)blockIdx.x ? c : 0;
  }
  for (int o = 32; o > 0; o >>= 1) { mine += __shfl_xor(mine, o); all += __shfl_xor(all, o); }
  if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = mine; wave_cnt[threadIdx.x >> 6] = all; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0, ta = 0;
    for (int w = 0; w < kCompactBlock / 64; ++w) { t += part[w]; ta += wave_cnt[w]; }
    base_sh = t;
    if (blockIdx.x == 0) *n_active = ta;
  }
  __syncthreads();
  const long ray = (long)blockIdx.x * kCompactBlock + threadIdx.x;
  const bool on = ray < n_rays && !(z_vals[ray * S + S - 1] == 0.f);
  const unsigned long long m = __ballot(on);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wave] = __popcll(m);
  __syncthreads();
  int off = base_sh;
  for (int w = 0; w < wave; ++w) off += wave_cnt[w];
  if (on) ray_index[off + __popcll(m & ((1ull << lane) - 1ull))] = (int)ray;
}

__global__ void points_in_boxes_kernel(const float* __restrict__ xyz, long n, const double* __restrict__ boxes,
                                       int n_boxes, uint8_t* __restrict__ inside) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  inside[idx] = in_any_box(xyz[idx * 3], xyz[idx * 3 + 1], xyz[idx * 3 + 2], boxes, n_boxes) ? 1 : 0;
}

// Ray generation for the editor (datasets/ray_utils.py:5-51, editable_renderer.py:153-181), as three device functions so
// that the one-kernel form (objnerf_generate_rays) and the stage-by-stage form the reference's caller is written
// against (get_ray_directions / get_rays / BBoxRayHelper.get_ray_bbox_intersections) are the same arithmetic.
// directions = [(i - W/2)/focal, -(j - H/2)/focal, -1]   (ray_utils.py:21-23; no +0.5)
__device__ __forceinline__ void pixel_direction(int x, int y, int W, int H, float focal, float dir[3]) {
  dir[0] = __fdiv_rn((float)x - (float)W / 2.f, focal);
  dir[1] = -__fdiv_rn((float)y - (float)H / 2.f, focal);
  dir[2] = -1.f;
}
// rays_d = directions @ c2w[:, :3].T, normalised (ray_utils.py:43-44); c2w row r at c2w + r * stride
__device__ __forceinline__ void rotate_normalise(const float dir[3], const float* c2w, int stride, float d[3]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) d[r] = dir[0] * c2w[r * stride + 0] + dir[1] * c2w[r * stride + 1] + dir[2] * c2w[r * stride + 2];
  const float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
#pragma unroll
  for (int r = 0; r < 3; ++r) d[r] = __fdiv_rn(d[r], nrm);
}
// float64 transform to box coordinates (bbox_utils.py:100-117), slab test (geo_utils.py:126-162) with the reference's
// rules: the direction is rotated by the box rotation only, zero components become 1e-14, an origin inside the box
// (tmin < 0) is a miss; near/far = entry/exit / scale_factor in fp32, 0/0 on a miss
__device__ __forceinline__ bool ray_box(const float o[3], const float d[3], const double* B, double enlarge, float& near, float& far) {
  const double sx = (double)o[0] * B[0], sy = (double)o[1] * B[0], sz = (double)o[2] * B[0];
  const double ax = B[1] * sx + B[2] * sy + B[3] * sz + B[10];
  const double ay = B[4] * sx + B[5] * sy + B[6] * sz + B[11];
  const double az = B[7] * sx + B[8] * sy + B[9] * sz + B[12];
  double ob[3], db[3];
  ob[0] = B[13] * ax + B[14] * ay + B[15] * az + B[22];
  ob[1] = B[16] * ax + B[17] * ay + B[18] * az + B[23];
  ob[2] = B[19] * ax + B[20] * ay + B[21] * az + B[24];
#pragma unroll
  for (int r = 0; r < 3; ++r)      // direction: box rotation only (bbox_utils.py:115)
    db[r] = B[13 + 3 * r] * (double)d[0] + B[14 + 3 * r] * (double)d[1] + B[15 + 3 * r] * (double)d[2];
  const double e = enlarge > 0 ? enlarge : 0.0;
  double tmin = 0, tmax = 0;
  bool hit = true;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double dd = db[r] == 0.0 ? 1.0e-14 : db[r];
    const double inv = 1.0 / dd;
    const double lo = B[25 + r] - e, hi = B[28 + r] + e;
    const double t0 = ((inv < 0 ? hi : lo) - ob[r]) * inv;
    const double t1 = ((inv < 0 ? lo : hi) - ob[r]) * inv;
    if (r == 0) { tmin = t0; tmax = t1; }
    else {
      if (tmin > t1 || t0 > tmax) hit = false;
      if (t0 > tmin) tmin = t0;
      if (t1 < tmax) tmax = t1;
    }
  }
  if (tmin < 0 || tmax < 0) hit = false;       // origin inside the box counts as a miss
  near = hit ? __fdiv_rn((float)tmin, (float)B[0]) : 0.f;
  far = hit ? __fdiv_rn((float)tmax, (float)B[0]) : 0.f;
  return hit;
}

struct GenRaysArgs {
  int H, W;
  float focal;
  float c2w[12];
  float near, far;
  int has_box;
  double box[OBJNERF_BOX_DOUBLES];
  double enlarge;
  // which image rows this call writes: local row lr -> image row row0 + (lr / row_block) * row_block * block_stride +
  // lr % row_block (contiguous band: row_block = n_rows, block_stride = 1; block-cyclic share of rank r of w:
  // row0 = r * row_block, block_stride = w)
  int row0, n_rows, row_block, block_stride;
};
__global__ void generate_rays_kernel(const GenRaysArgs a, float* __restrict__ rays) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)a.n_rows * a.W) return;
  const int lr = (int)(idx / a.W), x = (int)(idx - (long)lr * a.W);
  const int y = a.row0 + (lr / a.row_block) * a.row_block * a.block_stride + lr % a.row_block;
  float dir[3], d[3];
  pixel_direction(x, y, a.W, a.H, a.focal, dir);
  rotate_normalise(dir, a.c2w, 4, d);
  const float o[3] = {a.c2w[3], a.c2w[7], a.c2w[11]};
  float near = a.near, far = a.far;
  if (a.has_box) ray_box(o, d, a.box, a.enlarge, near, far);
  float* r = rays + idx * 8;
  r[0] = o[0]; r[1] = o[1]; r[2] = o[2]; r[3] = d[0]; r[4] = d[1]; r[5] = d[2]; r[6] = near; r[7] = far;
}

// the stage-by-stage form (dropin/datasets/ray_utils.py, dropin/utils/bbox_utils.py)
__global__ void ray_directions_kernel(int H, int W, float focal, float* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)H * W) return;
  const int y = (int)(idx / W), x = (int)(idx - (long)y * W);
  float dir[3];
  pixel_direction(x, y, W, H, focal, dir);
  out[idx * 3] = dir[0]; out[idx * 3 + 1] = dir[1]; out[idx * 3 + 2] = dir[2];
}
__global__ void get_rays_kernel(const float* __restrict__ directions, long n, const float* __restrict__ c2w, int stride,
                                float* __restrict__ rays_o, float* __restrict__ rays_d) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float dir[3] = {directions[idx * 3], directions[idx * 3 + 1], directions[idx * 3 + 2]};
  float d[3];
  rotate_normalise(dir, c2w, stride, d);
#pragma unroll
  for (int r = 0; r < 3; ++r) { rays_d[idx * 3 + r] = d[r]; rays_o[idx * 3 + r] = c2w[r * stride + 3]; }
}
struct BoxArg { double box[OBJNERF_BOX_DOUBLES]; double enlarge; };
__global__ void ray_box_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, long n, const BoxArg b,
                               uint8_t* __restrict__ hit, float* __restrict__ near, float* __restrict__ far) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float o[3] = {rays_o[idx * 3], rays_o[idx * 3 + 1], rays_o[idx * 3 + 2]};
  const float d[3] = {rays_d[idx * 3], rays_d[idx * 3 + 1], rays_d[idx * 3 + 2]};
  float nr, fr;
  const bool h = ray_box(o, d, b.box, b.enlarge, nr, fr);
  hit[idx] = h ? 1 : 0; near[idx] = nr; far[idx] = fr;
}

// ------------------------------------------------------------------------------------------
// Per-ray vectors of the hoisted terms (mlp_kernel HOIST).  For the layers whose input is cat([.., x]) with x constant
// along a ray -- the object code in instance_encoding_1 / _3 (nerf_model.py:128-138), the direction embedding in
// dir_encoding / inst_dir_encoding (116, 147) -- out[ray] = bias + W[:, columns of x] . x, written in the layer's
// D-register order so that a lane of the MLP kernel reads its 16 values as one 64-byte piece.
// Two steps: the weight columns are gathered into an MFMA A-operand stream when the weights are packed (pack_all_kernel), then
// ray_bias_kernel forms the product on the matrix pipe (round 6; rounds 3-5: lane = ray on the VALU with wave-uniform weight
// operands, 353 us per frame pass; round 3 first tried thread = output row with the rays' inputs broadcast from LDS: 1.07 ms).
// ------------------------------------------------------------------------------------------
struct RayBiasArgs {
  const float* blob; const float* aux; const float* rays; const float* codes;
  long code_stride, n_rays;
  int use_voxel, do_scene, do_object;
  float* out;
  const int32_t* ray_index; const int32_t* n_active;      // optional ray subset (objnerf_mlp_args): only the listed rays are computed
};
// position in the packed weight stream of W[row][column held by k-step `ks`, lane half `half`] of layer l (layout.h)
__device__ __forceinline__ long blob_weight_index(bool vox, int l, int ks, int half, int row) {
  const int nt = layer_nt(l), kg = kChunkTiles / nt;
  const long base = (long)layer_chunk_start(vox, l) * kChunkFloats;
  const int chunk = ks / kg, kl = ks % kg, g4 = kl / 4, j = kl % 4, m = row >> 5, lane = (row & 31) + 32 * half;
  return base + (long)chunk * kChunkFloats + ((long)(g4 * nt + m) * 64 + lane) * 4 + j;
}
// Step 1 (inside the packer, pack_all_kernel): the hoisted weight columns as a compact matrix wm[group][c][16] + bias[group][16],
// group = 16 consecutive floats of the per-ray vector (one (layer, out tile, lane half) of the MLP kernel's D layout), c = input
// column (64 code columns or 27 direction columns).  This returns WHERE element (o, c) sits in the packed stream (-1: zero) by the
// packer's own layout arithmetic; the packer resolves it through the same gather map the stream itself is written through.
__device__ __forceinline__ long rb_matrix_src(bool vox, int o, int c) {
  const int l = o < 128 ? L_O1 : (o < 256 ? L_O3 : (o < 384 ? L_SD : L_OD));
  const int off = o < 128 ? 0 : (o < 256 ? 128 : (o < 384 ? 256 : 384));
  const int q = o - off, m = q >> 5, half = (q >> 4) & 1, r = q & 15;
  const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * half;   // D layout
  if (o < 256) {
    const int ks0 = ks_emb(vox) + (vox ? kKsObjVox : 0);        // first code k-step of the object input list (layout.h)
    return blob_weight_index(vox, l, ks0 + (c & 31), c >> 5, row);
  }
  long src = -1;
  if (c < kDirC) {
    const int nh = l == L_SD ? 128 : 64;
    // slot (i, h) of the direction list that holds column c (layout.h::dir_slot_col)
    for (int i = 0; i < kKsDir; ++i)
      for (int h = 0; h < 2; ++h)
        if (dir_slot_col(i, h) == c) src = blob_weight_index(vox, l, nh + i, h, row);
  }
  return src;
}
// aux position of the bias that starts position o of the per-ray vector
__device__ __forceinline__ int rb_bias_src(int o) {
  const int l = o < 128 ? L_O1 : (o < 256 ? L_O3 : (o < 384 ? L_SD : L_OD));
  const int off = o < 128 ? 0 : (o < 256 ? 128 : (o < 384 ? 256 : 384));
  return aux_bias_off(l) + (o - off);
}
// Step 2 (round 6): out^T[448, rays] = A[448, 91] X^T[91, rays] on the matrix pipe.  A workgroup (4 waves) takes 32 rays at a
// time; wave w owns the out tiles {2w, 2w+1} of the code product and {8+2w, 9+2w} (w < 2) / {12 + w - 2} (w >= 2) of the
// direction product -- 92 / 92 / 78 / 78 MFMAs -- and keeps ITS tiles' A operands in its own slice of LDS (staged once per
// workgroup, no block-wide barrier: a wave only reads what it wrote).  Lane (ray, h) holds the B operands in registers: the 32
// code floats 32 h .. 32 h + 31 and the 28 direction-embedding values (the MLP kernel's own sin / cos).  The D layout of a tile
// IS the 16-float piece [m][h][0..15] of the ray's vector: four 16-byte stores per lane and tile.  Rounds 3-5 computed the same
// sums lane = ray on the VALU (29 k FMAs per ray, 353 us per frame pass at 0.9 of the plain-FMA rate).
constexpr int kRbWaveFloats = (2 * kRbCodeQ + 2 * kRbDirQ) * 256;       // the largest per-wave A slice: 24 KB
__device__ __forceinline__ f32x4 rb_lds16(const float* p) { return *(const __attribute__((address_space(3))) f32x4*)p; }

constexpr int kRbPatchLd = 36;                       // floats per ray row of a wave's store patch (32 + 4: rows land on distinct banks)
template <int NT, int NQ, int KS>
__device__ __forceinline__ void rb_tiles(const float* a_lds, const float* bias_lds, const int (&off)[2], int lane, float* patch,
                                         const int (&rayrow)[4], float* out_base, const float (&bop)[KS]) {
  // NT tiles side by side (independent accumulators between dependent MFMAs), NQ groups of 4 k-steps, KS valid k-steps
  f32x16 acc[NT];
  const int h = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const f32x4 b = rb_lds16(bias_lds + off[t] + 16 * h + 4 * q4);
      acc[t][4 * q4] = b[0]; acc[t][4 * q4 + 1] = b[1]; acc[t][4 * q4 + 2] = b[2]; acc[t][4 * q4 + 3] = b[3];
    }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    f32x4 a4[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) a4[t] = rb_lds16(a_lds + ((t * NQ + q) * 64 + lane) * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * q + j < KS) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t][j], bop[4 * q + j], acc[t], 0, 0, 0);
      }
  }
  // The D layout of a tile is the 16-float piece [m][h][0..15] of the ray's vector: 64 bytes per lane, a ray's two halves 128
  // contiguous bytes.  Stored straight from the registers, an instruction writes 16 bytes into each of 64 different 64-byte
  // segments -- 34 M quarter-filled write requests per frame pass, which is what the kernel then waits for (probes,
  // profiles/r06_ray_bias_probe.txt: 253 us with the stores, 139 without, 125 without stores and input loads).  Through the wave's
  // own LDS patch 8 consecutive lanes write one ray's whole 128-byte line (a quarter of the requests).
  // EVERY lane stores, no branch around the stores: a row past the end of the batch carries the LAST ray's inputs (fetch clamps its
  // slot) and re-writes that ray's row with the same bits.  A `valid` predicate made the stores a skippable block, the wait-count
  // pass then priced every wait for the next group's inputs by the path without them: vmcnt(0), a full drain per iteration.
  const int pt = lane & 31, rq = lane >> 3, k = lane & 7;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4)
      *(f32x4*)(patch + pt * kRbPatchLd + 16 * h + 4 * q4) = f32x4{acc[t][4 * q4], acc[t][4 * q4 + 1], acc[t][4 * q4 + 2], acc[t][4 * q4 + 3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 v = *(const f32x4*)(patch + (8 * i + rq) * kRbPatchLd + 4 * k);
      *(f32x4u*)(out_base + (long)rayrow[i] * kRayBiasFloats + off[t] + 4 * k) = v;
    }
  }
}

// Eight waves: the A slices (96 KB: one workgroup per CU) leave room for only ONE wave per SIMD with four -- nothing then overlaps a
// wave's own sin / cos, operand selects, stores and input round trips with its MFMAs (matrix pipe 0.35 busy, 235 us).  Waves w and
// w + 4 share tile set w's slice and take alternate 32-ray groups: two waves per SIMD, one's MFMAs under the other's VALU / memory.
// SCENE / OBJ are template parameters so that the usual call (both) has no uniform branch around the input loads or the stores: the
// wait-count pass prices a wait by the path with the FEWEST memory operations behind the awaited one, and a skippable block of
// stores or loads between a load and its use turns the wait into vmcnt(0).
template <bool SCENE, bool OBJ>
__global__ void __launch_bounds__(512) ray_bias_kernel(const RayBiasArgs a, const float* __restrict__ rbA) {
  __shared__ __attribute__((aligned(16))) float a_lds[4 * kRbWaveFloats];
  __shared__ __attribute__((aligned(16))) float bias_lds[kRayBiasFloats];
  __shared__ __attribute__((aligned(16))) float patch_lds[8 * 32 * kRbPatchLd];       // a 32-ray x 128-byte store patch per wave
  const long n = a.n_active ? (long)*a.n_active : a.n_rays;
  const long groups = (n + 31) >> 5;
  if (2 * (long)blockIdx.x >= groups) return;                   // (uniform) the grid is sized for n_rays, a culled subset may need less
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((tid >> 6) & 3), sub = __builtin_amdgcn_readfirstlane(tid >> 8);
  // The two waves of a SIMD do identical work from an identical start: at equal priority they share the matrix pipe round-robin,
  // finish their MFMAs together and then store together -- chip-wide, product phases and store phases alternate instead of
  // overlapping (probes: 144 us of products, 138 us of stores at the write path's 4 TB/s, 225 us together).  With set 0 above set 1
  // the first runs its products at full rate while the second waits, then stores while the second computes: anti-phase from then on.
  if (sub == 0) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(0);
  // this wave's tiles
  const int tc = 2 * wave;                                      // code tiles tc, tc + 1
  const int td = wave < 2 ? kRbCodeTiles + 2 * wave : kRbCodeTiles + 2 + wave;      // direction tiles td (, td + 1 when wave < 2)
  const int nd = wave < 2 ? 2 : 1;
  constexpr bool code_live = OBJ;
  const bool dir_live = (SCENE && OBJ) ? true : (td < 12 ? SCENE : OBJ);
  float* my = a_lds + wave * kRbWaveFloats;
  {
    // stage: [code tile tc | code tile tc + 1 | direction tile(s)] -- contiguous runs of the stream, 16 bytes per lane and step
    const float* src_c = rbA + rb_tile_start(tc);
    const float* src_d = rbA + rb_tile_start(td);
    const int n_c = 2 * kRbCodeQ * 64, n_d = nd * kRbDirQ * 64;           // float4 counts
    if (code_live)
      for (int i = lane + 64 * sub; i < n_c; i += 128) *(f32x4*)(my + 4 * i) = gload4(src_c + 4 * i);
    if (dir_live)
      for (int i = lane + 64 * sub; i < n_d; i += 128) *(f32x4*)(my + 2 * kRbCodeQ * 256 + 4 * i) = gload4(src_d + 4 * i);
    for (int i = tid; i < kRayBiasFloats; i += 512) bias_lds[i] = rbA[kRbAFloats + i];
  }
  __syncthreads();                                              // (the biases are shared; an A slice belongs to one pair of waves)
  const int off_c[2] = {rb_tile_off(tc), rb_tile_off(tc + 1)};
  const int off_d[2] = {rb_tile_off(td), rb_tile_off(td + (nd == 2 ? 1 : 0))};
  // the inputs of a 32-ray group: this lane's half of the ray's code and the ray's direction; the NEXT group's are in flight
  // while the current group's products run (one wave per SIMD: nothing else hides the round trip)
  struct In { float x[32]; float d[3]; long ray; };
  auto fetch = [&](long grp, In& o) __attribute__((always_inline)) {
    const long slot = grp * 32 + (lane & 31);
    const long sl = slot < n ? slot : n - 1;                 // lanes past the end repeat the last ray (and re-write its row, see rb_tiles)
    o.ray = a.ray_index ? (long)a.ray_index[sl] : sl;        // vectors stay indexed by the ray's own number
    if (code_live) {
      const float* cp = a.codes + o.ray * a.code_stride + 32 * h;
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4) {
        const f32x4 v = *(const f32x4u*)(cp + 4 * c4);
        o.x[4 * c4] = v[0]; o.x[4 * c4 + 1] = v[1]; o.x[4 * c4 + 2] = v[2]; o.x[4 * c4 + 3] = v[3];
      }
    }
    if (dir_live) {
#pragma unroll
      for (int c = 0; c < 3; ++c) o.d[c] = a.rays[o.ray * 8 + 3 + c];
    }
  };
  float* const patch = patch_lds + (tid >> 6) * 32 * kRbPatchLd;
  auto compute = [&](const In& in) __attribute__((always_inline)) {
    // the ray numbers of the four rows this lane stores for (row 8 i + lane / 8 of the group: lane `row` holds it)
    int rayrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rayrow[i] = __shfl((int)in.ray, 8 * i + (lane >> 3));
    float* const out = a.out;
    if (code_live) rb_tiles<2, kRbCodeQ, 32>(my, bias_lds, off_c, lane, patch, rayrow, out, in.x);
    if (dir_live) {
      // Embedding(3, 4): [d, sin(2^k d), cos(2^k d)] (embedding_helper.py:69-74), the MLP kernel's own sin / cos.  Lane half h feeds
      // columns 14 h .. 14 h + 13 of the 27 (+ one zero): half 0 = [d | sin, cos of 2^0 d | sin 2 d | cos 2 d (x, y)], half 1 =
      // [cos 2 d (z) | sin, cos of 4 d | sin, cos of 8 d | 0] -- so a lane evaluates the two octaves of ITS half (argument 2^(2h) d
      // and twice that) plus cos 2 d_z: 7 sin/cos pairs instead of all 12
      const float f0 = h ? 4.f : 1.f;
      float s0[3], c0[3], s1[3], c1[3];
#pragma unroll
      for (int coord = 0; coord < 3; ++coord) {
        const SinCos a0 = psincos(in.d[coord] * f0), a1 = psincos(in.d[coord] * (2.f * f0));
        s0[coord] = a0.s; c0[coord] = a0.c; s1[coord] = a1.s; c1[coord] = a1.c;
      }
      const float cz = psincos(in.d[2] * 2.f).c;                 // column 14 = cos(2 d_z)
      float b[kRbDirKs];
      b[0] = h ? cz : in.d[0];  b[1] = h ? s0[0] : in.d[1]; b[2] = h ? s0[1] : in.d[2];
      b[3] = h ? s0[2] : s0[0]; b[4] = h ? c0[0] : s0[1];  b[5] = h ? c0[1] : s0[2];
      b[6] = h ? c0[2] : c0[0]; b[7] = h ? s1[0] : c0[1];  b[8] = h ? s1[1] : c0[2];
      b[9] = h ? s1[2] : s1[0]; b[10] = h ? c1[0] : s1[1]; b[11] = h ? c1[1] : s1[2];
      b[12] = h ? c1[2] : c1[0]; b[13] = h ? 0.f : c1[1];
      const float* ad = my + 2 * kRbCodeQ * 256;
      if (nd == 2) rb_tiles<2, kRbDirQ, kRbDirKs>(ad, bias_lds, off_d, lane, patch, rayrow, out, b);
      else rb_tiles<1, kRbDirQ, kRbDirKs>(ad, bias_lds, off_d, lane, patch, rayrow, out, b);
    }
  };
  // Two input sets, used alternately (no register copies at the loop end: the copies of a single-buffered form were scheduled right
  // behind the loads, each with its own vmcnt(0)): the inputs of the group after this one are requested before this group's products
  // and stores, and first read one whole iteration later.  (Three sets -- the awaited loads then sit behind stores two iterations
  // old instead of one, vmcnt counting loads and stores in issue order -- measured the same 220 us and spilled: two stay.)
  const long stride = 2 * (long)gridDim.x;
  In i0, i1;
  long grp = 2 * (long)blockIdx.x + sub;
  if (grp >= groups) return;                                 // (wave-uniform; no barrier follows)
  fetch(grp, i0);
  // the first group's inputs are waited for HERE: entering the loop with them in flight, the loop's own waits for set 0 would be
  // priced by this entry path (8 younger operations) instead of by the back edge (~20 stores younger) on every iteration
  __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0)
  // (the fetch is unconditional -- past the last group every lane clamps to the last ray: a skipped fetch made the two sets merge
  // through register copies placed right behind the loads, each copy waiting with vmcnt(0))
  while (true) {
    fetch(grp + stride, i1);
    compute(i0);
    grp += stride;
    if (grp >= groups) break;
    fetch(grp + stride, i0);
    compute(i1);
    grp += stride;
    if (grp >= groups) break;
  }
}

// volume_rendering_multi (multi_rendering.py:96-157): joint stable sort by z of K*S samples,
// gather, composite with last delta 0.  One wave per ray, everything staged in LDS.
// Staging area per ray = 7 floats per sample (28 B): in LDS up to kMultiLdsMax bytes (M = K*S <= 5,558 samples; beyond the
// default 64 KiB of dynamic LDS the launch raises the kernel's limit once), in a caller-provided global scratch beyond
// that (GLOBAL: one slice per workgroup, L2-resident; the reference sorts any K*S, multi_rendering.py:96-157).
constexpr int kMaxSets = 64;
constexpr size_t kMultiLdsMax = 152 * 1024;
constexpr unsigned kMultiGlobalGrid = 1024;      // workgroups (= scratch slices) of the global-staging variant
struct MultiPtrs {
  const float* z[kMaxSets];
  const float* sigma[kMaxSets];
  const float* rgb[kMaxSets];
  float* own_w[kMaxSets];
};

template <bool GLOBAL>
__global__ void __launch_bounds__(64) composite_multi_kernel(const MultiPtrs ptrs, long n_rays, int K, int S,
                                                             const float* __restrict__ noise, float noise_std,
                                                             int white_back, float* __restrict__ z_sorted,
                                                             float* __restrict__ weights, float* __restrict__ obj_ids,
                                                             float* __restrict__ opacity, float* __restrict__ rgb_map,
                                                             float* __restrict__ depth, int has_own, float* stage) {
  extern __shared__ __attribute__((aligned(16))) float sm_lds[];
  const int M = K * S;
  float* sm = GLOBAL ? stage + (size_t)blockIdx.x * 7 * M : sm_lds;
  float* zin = sm;            // M   unsorted z
  float* zs = sm + M;         // M   sorted z
  float* sgs = sm + 2 * M;    // M   sorted sigma
  float* cs = sm + 3 * M;     // 3M  sorted rgb
  int* rk = (int*)(sm + 6 * M);   // M  rank of unsorted element
  const int lane = threadIdx.x;
  for (long ray = blockIdx.x; ray < n_rays; ray += gridDim.x) {
    bool asc = true;
    for (int i = lane; i < M; i += 64) {
      const int s = i % S;
      const float* zk = ptrs.z[i / S] + ray * S;
      const float v = zk[s];
      zin[i] = v;
      asc = asc && (s == 0 || !(v < zk[s - 1]));       // every set's depths ascending (NaN counts as not)
      asc = asc && v == v;
    }
    asc = __ballot(asc) == ~0ull;
    __syncthreads();
    for (int i = lane; i < M; i += 64) {
      const float v = zin[i];
      const int k = i / S, s = i % S;
      int rank = 0;
      if (asc) {
        // K ascending runs: the stable rank (ties in concatenation order, as torch.sort(..., stable) would give and as
        // the O(M^2) count below defines it) is the element's position in its own run plus, per other run, the number
        // of its elements that go in front: <= v for the runs before it, < v for the runs behind it -- K binary
        // searches of log2(S) steps instead of M comparisons
        rank = s;
        for (int kk = 0; kk < K; ++kk)
          if (kk != k) rank += count_below(zin + kk * S, S, v, kk < k);
      } else {
        for (int j = 0; j < M; ++j) {
          const float o = zin[j];
          rank += (o < v || (o == v && j < i)) ? 1 : 0;
        }
      }
      rk[i] = rank;
      zs[rank] = v;
      sgs[rank] = ptrs.sigma[k][ray * S + s];
      const float* c = ptrs.rgb[k] + (ray * S + s) * 3;
      cs[rank * 3] = c[0]; cs[rank * 3 + 1] = c[1]; cs[rank * 3 + 2] = c[2];
      if (obj_ids) obj_ids[ray * M + rank] = (float)k;
    }
    __syncthreads();
    float* wrow = weights + ray * M;
    const CompositeOut o = composite_ray(zs, sgs, cs, noise ? noise + ray * M : nullptr, noise_std,
                                         0.f, M, lane, false, 0.f, wrow);
    for (int i = lane; i < M; i += 64) z_sorted[ray * M + i] = zs[i];
    if (lane == 0) {
      opacity[ray] = o.opacity;
      depth[ray] = o.depth;
      rgb_map[ray * 3 + 0] = white_back ? o.r + 1.f - o.opacity : o.r;
      rgb_map[ray * 3 + 1] = white_back ? o.g + 1.f - o.opacity : o.g;
      rgb_map[ray * 3 + 2] = white_back ? o.b + 1.f - o.opacity : o.b;
    }
    if (has_own) {
      // weights[obj_ids == k] in own sample order (multi_rendering.py:269-271); same wave wrote wrow
      __syncthreads();
      for (int i = lane; i < M; i += 64) ptrs.own_w[i / S][ray * S + (i % S)] = wrow[rk[i]];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// weight packer: gather through the index map
// ------------------------------------------------------------------------------------------
struct ParamPtrs { const float* p[kNumParamPtrs]; };
__device__ __forceinline__ float pack_fetch(const ParamPtrs& pp, uint32_t e) {
  return e == kPackZero ? 0.f : pp.p[e >> 24][e & 0xFFFFFFu];
}
__global__ void pack_kernel(const uint32_t* __restrict__ idx, long n, const ParamPtrs pp, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = pack_fetch(pp, idx[i]);
}
// Everything the inference kernels read of up to kPackModels ObjectNeRF modules in ONE launch (round 6: the product re-gathers on
// EVERY call instead of caching per parameter version -- a cache keyed on host-visible versions cannot see `.data` writes, fused
// optimizers or graph-replayed steps; three launches per model were ~40 us per render_rays call, this is one of ~8 us).
// blockIdx.y = model; blockIdx.x walks [weight stream | aux block | compact matrix of the hoisted columns + its biases]; the
// compact matrix reads the PARAMETERS through the gather maps (not the freshly packed stream: no ordering inside one launch).
constexpr int kPackModels = 2;
struct PackJob { ParamPtrs pp; float* blob; float* aux; };
struct PackJobs { PackJob j[kPackModels]; };
__global__ void __launch_bounds__(256) pack_all_kernel(const uint32_t* __restrict__ blob_idx, long nb, const uint32_t* __restrict__ aux_idx,
                                                       long na, int use_voxel, const PackJobs jobs) {
  const PackJob& job = jobs.j[blockIdx.y];
  // four consecutive stream floats per thread: one 16-byte read of the gather map, four gathers, one 16-byte store (the stream
  // length is a multiple of the chunk size; the map and the stream come from the caching allocator: 16-byte aligned)
  const long nbB = (nb / 4 + 255) / 256, naB = (na + 255) / 256;
  long b = blockIdx.x;
  if (b < nbB) {
    const long i = b * 256 + threadIdx.x;
    if (i < nb / 4) {
      const uint4 e = *(const uint4*)(blob_idx + 4 * i);
      f32x4 v;
      v[0] = pack_fetch(job.pp, e.x); v[1] = pack_fetch(job.pp, e.y); v[2] = pack_fetch(job.pp, e.z); v[3] = pack_fetch(job.pp, e.w);
      *(f32x4*)(job.blob + 4 * i) = v;
    }
    return;
  }
  b -= nbB;
  if (b < naB) {
    const long i = b * 256 + threadIdx.x;
    if (i < na) job.aux[i] = pack_fetch(job.pp, aux_idx[i]);
    return;
  }
  b -= naB;
  const long t = b * 256 + threadIdx.x;
  float* rbA = job.aux + kAuxFloats;
  if (t < kRbAFloats) {
    // element (T, q, lane, j) of ray_bias_kernel's A stream (layout.h): row 32 m + (lane & 31) of tile T, k-step 4 q + j, half h
    const int blk = (int)(t >> 8), lane = (int)(t >> 2) & 63, j = (int)t & 3;
    const bool code = blk < kRbCodeTiles * kRbCodeQ;
    const int T = code ? blk / kRbCodeQ : kRbCodeTiles + (blk - kRbCodeTiles * kRbCodeQ) / kRbDirQ;
    const int q = code ? blk % kRbCodeQ : (blk - kRbCodeTiles * kRbCodeQ) % kRbDirQ;
    const int ks = 4 * q + j, h = lane >> 5, row = lane & 31;
    const int c = code ? 32 * h + ks : (ks < kRbDirKs ? kRbDirKs * h + ks : 64);
    // row of the tile -> its place in the per-ray vector (the D layout: row = (r & 3) + 8 (r >> 2) + 4 half)
    const int o = rb_tile_off(T) + 16 * ((row >> 2) & 1) + (row & 3) + 4 * (row >> 3);
    const long src = (code || c < kDirC) ? rb_matrix_src(use_voxel != 0, o, c) : -1;
    rbA[t] = src < 0 ? 0.f : pack_fetch(job.pp, blob_idx[src]);
  } else if (t < kRbMatFloats) {
    const int o = (int)(t - kRbAFloats);
    rbA[t] = pack_fetch(job.pp, aux_idx[rb_bias_src(o)]);
  }
}

}  // namespace objnerf

// ==========================================================================================
// host side of the C ABI for these stages
// ==========================================================================================
using namespace objnerf;

static inline unsigned blocks_for(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

extern "C" {

int objnerf_sample_coarse(const float* rays, const float* z_steps, const float* perturb_rand, float perturb,
                          int use_disp, int64_t n_rays, int S, float* z_vals, void* stream) {
  if (!rays || !z_steps || !z_vals || S < 1 || n_rays < 0) return set_error(-1, "sample_coarse: bad arguments");
  if (perturb > 0.f && !perturb_rand) return set_error(-1, "sample_coarse: perturb > 0 needs perturb_rand");
  if (n_rays == 0) return 0;
  if (!(perturb > 0.f) && (S & 3) == 0 && (((uintptr_t)z_steps | (uintptr_t)z_vals) & 15) == 0) {
    hipLaunchKernelGGL(sample_coarse4_kernel, dim3(blocks_for(n_rays * (S >> 2), 256)), dim3(256), 0, (hipStream_t)stream,
                       rays, z_steps, use_disp, (long)n_rays, S, z_vals);
    return check_launch("sample_coarse");
  }
  hipLaunchKernelGGL(sample_coarse_kernel, dim3(blocks_for(n_rays * S, 256)), dim3(256), 0, (hipStream_t)stream,
                     rays, z_steps, perturb_rand, perturb, use_disp, (long)n_rays, S, z_vals);
  return check_launch("sample_coarse");
}

int objnerf_pos_encode(const float* x, int64_t n, int C, int n_freqs, float* out, void* stream) {
  return objnerf_pos_encode_freqs(x, n, C, n_freqs, nullptr, out, stream);
}
int objnerf_pos_encode_freqs(const float* x, int64_t n, int C, int n_freqs, const float* freqs, float* out, void* stream) {
  if (!x || !out || C < 1 || n_freqs < 0) return set_error(-1, "pos_encode: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(pos_encode_kernel, dim3(blocks_for(n * C, 256)), dim3(256), 0, (hipStream_t)stream,
                     x, (long)n, C, n_freqs, freqs, out);
  return check_launch("pos_encode");
}

int objnerf_voxel_embed(const objnerf_voxel_grid* grid, const float* xyz, int64_t n, float* scene_ftr,
                        float* obj_ftr, void* stream) {
  if (!grid || !grid->idx_map || !grid->table || !xyz || !scene_ftr || !obj_ftr)
    return set_error(-1, "voxel_embed: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(voxel_embed_kernel, dim3(blocks_for(n, 8)), dim3(256), 0, (hipStream_t)stream,
                     *grid, xyz, (long)n, scene_ftr, obj_ftr);
  return check_launch("voxel_embed");
}

int objnerf_composite(const objnerf_composite_args* a, void* stream) {
  if (!a || !a->z_vals || !a->sigma || !a->rgb || !a->weights || !a->opacity || !a->rgb_map || !a->depth)
    return set_error(-1, "composite: bad arguments");
  if (a->inst_sigma && (!a->inst_rgb || !a->rgb_inst || !a->depth_inst || !a->opacity_inst))
    return set_error(-1, "composite: instance outputs missing");
  if (a->noise_std != 0.f && (!a->noise || (a->inst_sigma && !a->noise_inst)))
    return set_error(-1, "composite: noise_std != 0 needs noise draws");
  if (a->n_rays == 0) return 0;
  objnerf_composite_args c = *a;
  if (c.noise_std == 0.f) { c.noise = nullptr; c.noise_inst = nullptr; }
  const long waves = c.n_rays;
  unsigned grid = (unsigned)((waves + 3) / 4);
  if (grid > 256u * 32u) grid = 256u * 32u;
  hipLaunchKernelGGL(composite_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, c);
  return check_launch("composite");
}

// K4 keeps one wave per ray; as many rays per workgroup (1, 2 or 4) as fit the default 64 KiB of dynamic LDS
static inline int waves_per_block(size_t lds_per_wave) {
  return lds_per_wave * 4 <= 48 * 1024 ? 4 : (lds_per_wave * 2 <= 48 * 1024 ? 2 : 1);
}

int objnerf_sample_pdf(const float* bins, const float* weights, const float* u, int64_t u_stride, int64_t n_rays,
                       int nb, int I, float eps, float* samples, void* stream) {
  if (!bins || !weights || !u || !samples || nb < 2 || nb > kMaxBins || I < 1)
    return set_error(-1, "sample_pdf: bad arguments (2 <= bins <= 1024)");
  if (n_rays == 0) return 0;
  const size_t per_wave = (size_t)2 * nb * sizeof(float);
  const int nwb = waves_per_block(per_wave);
  const long blocks = (n_rays + nwb - 1) / nwb;
  unsigned grid = (unsigned)(blocks < 65536 ? blocks : 65536);
  hipLaunchKernelGGL(sample_pdf_kernel, dim3(grid), dim3(64 * nwb), per_wave * nwb, (hipStream_t)stream, bins,
                     weights, u, (long)u_stride, (long)n_rays, nb, I, eps, samples);
  return check_launch("sample_pdf");
}

int objnerf_sample_pdf_merge(const float* z_coarse, const float* weights, const float* u, int64_t u_stride,
                             int64_t n_rays, int S, int I, float eps, float* z_samples, float* z_fine,
                             void* stream) {
  return objnerf_sample_pdf_merge_clip(z_coarse, weights, u, u_stride, n_rays, S, I, eps, z_samples, z_fine, nullptr, stream);
}

int objnerf_sample_pdf_merge_clip(const float* z_coarse, const float* weights, const float* u, int64_t u_stride,
                                  int64_t n_rays, int S, int I, float eps, float* z_samples, float* z_fine,
                                  const float* clip, void* stream) {
  if (!z_coarse || !weights || !u || !z_fine || S < 3 || S - 1 > kMaxBins || I < 1 || S + I > kMaxMerge)
    return set_error(-1, "sample_pdf_merge: bad arguments (3 <= S <= 1025, S + I <= 2048)");
  if (n_rays == 0) return 0;
  if (S == 64 && (I == 64 || I == 128)) {           // the shipped sample counts: the specialised, persistent form
    const long blocks = (n_rays + 3) / 4;
    const unsigned grid = (unsigned)(blocks < 2048 ? blocks : 2048);           // 8 workgroups of 4 waves per CU
    if (I == 64)
      hipLaunchKernelGGL(sample_pdf_merge64_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, z_coarse, weights, u, (long)u_stride,
                         (long)n_rays, eps, z_samples, z_fine, clip);
    else
      hipLaunchKernelGGL(sample_pdf_merge64_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, z_coarse, weights, u, (long)u_stride,
                         (long)n_rays, eps, z_samples, z_fine, clip);
    return check_launch("sample_pdf_merge");
  }
  const size_t per_wave = (size_t)(2 * (S - 1) + (S + I) + I) * sizeof(float);      // <= 6142 floats = 24 KiB
  const int nwb = waves_per_block(per_wave);
  const long blocks = (n_rays + nwb - 1) / nwb;
  unsigned grid = (unsigned)(blocks < 65536 ? blocks : 65536);
  hipLaunchKernelGGL(sample_pdf_merge_kernel, dim3(grid), dim3(64 * nwb), per_wave * nwb, (hipStream_t)stream, z_coarse,
                     weights, u, (long)u_stride, (long)n_rays, S, I, eps, z_samples, z_fine, clip);
  return check_launch("sample_pdf_merge");
}

int objnerf_mask_sigma(float* sigma, const float* rays, const float* z_vals, int64_t n_rays, int S,
                       const double* boxes, int n_boxes, void* stream) {
  return objnerf_mask_sigma_rgb(sigma, nullptr, rays, z_vals, n_rays, S, boxes, n_boxes, stream);
}

int objnerf_mask_sigma_rgb(float* sigma, float* rgb, const float* rays, const float* z_vals, int64_t n_rays, int S,
                           const double* boxes, int n_boxes, void* stream) {
  if (!sigma || !rays || !z_vals || S < 1 || (n_boxes > 0 && !boxes)) return set_error(-1, "mask_sigma: bad arguments");
  if (n_rays == 0) return 0;
  hipLaunchKernelGGL(mask_sigma_kernel, dim3(blocks_for(n_rays * S, 256)), dim3(256), 0, (hipStream_t)stream,
                     sigma, rgb, rays, z_vals, (long)n_rays, S, boxes, n_boxes);
  return check_launch("mask_sigma");
}

int64_t objnerf_compact_scratch_ints(int64_t n_rays) { return (n_rays + kCompactBlock - 1) / kCompactBlock + 1; }

int objnerf_compact_rays(const float* z_vals, int64_t n_rays, int S, int32_t* ray_index, int32_t* n_active,
                         int32_t* scratch, void* stream) {
  if (!z_vals || !ray_index || !n_active || !scratch || S < 1 || n_rays < 0 || n_rays > 0x7fffffffll)
    return set_error(-1, "compact_rays: bad arguments");
  const unsigned nb = blocks_for(n_rays, kCompactBlock);
  if (nb == 0) return hipMemsetAsync(n_active, 0, sizeof(int32_t), (hipStream_t)stream) == hipSuccess ? 0 : set_error(-2, "compact_rays: memset failed");
  hipLaunchKernelGGL(ray_count_kernel, dim3(nb), dim3(kCompactBlock), 0, (hipStream_t)stream, z_vals, (long)n_rays, S, scratch);
  hipLaunchKernelGGL(ray_compact_kernel, dim3(nb), dim3(kCompactBlock), 0, (hipStream_t)stream, z_vals, (long)n_rays, S,
                     scratch, ray_index, n_active);
  return check_launch("compact_rays");
}

int objnerf_points_in_boxes(const float* xyz, int64_t n, const double* boxes, int n_boxes, uint8_t* inside,
                            void* stream) {
  if (!xyz || !inside || (n_boxes > 0 && !boxes)) return set_error(-1, "points_in_boxes: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(points_in_boxes_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     xyz, (long)n, boxes, n_boxes, inside);
  return check_launch("points_in_boxes");
}

int objnerf_composite_finish(const float* seg_records, int64_t n_rays, int S, int has_instance, int inst_weights,
                             int white_back, float* weights, float* opacity, float* rgb_map, float* depth, float* rgb_inst,
                             float* depth_inst, float* opacity_inst, void* stream) {
  if (n_rays < 0 || S < 32 || (S & 31)) return set_error(-1, "composite_finish: S must be a positive multiple of 32");
  if (n_rays == 0) return 0;
  if (!seg_records || !weights || !opacity || !rgb_map || !depth) return set_error(-1, "composite_finish: null pointer");
  if (has_instance && (!rgb_inst || !depth_inst || !opacity_inst)) return set_error(-1, "composite_finish: instance outputs missing");
  if (inst_weights && !has_instance) return set_error(-1, "composite_finish: inst_weights without the instance set");
  // OBJNERF_FINISH=wave: the one-wave-per-ray form of round 3 (A/B switch; results are bit-equal)
  static const bool per_wave = [] { const char* e = getenv("OBJNERF_FINISH"); return e && !strcmp(e, "wave"); }();
  const long blocks = (n_rays + kFinishRays - 1) / kFinishRays;
  if (per_wave || blocks > 0x7fffffffL) {
    const long waves = n_rays < 262144 ? n_rays : 262144;
    hipLaunchKernelGGL(composite_finish_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, seg_records,
                       (long)n_rays, S, has_instance != 0, white_back, weights, opacity, rgb_map, depth, rgb_inst, depth_inst,
                       opacity_inst, inst_weights != 0);
  } else {
    hipLaunchKernelGGL(composite_finish_block_kernel, dim3((unsigned)blocks), dim3(256), (size_t)kFinishRays * (S >> 5) * sizeof(float),
                       (hipStream_t)stream, seg_records, (long)n_rays, S, has_instance != 0, white_back, weights, opacity, rgb_map,
                       depth, rgb_inst, depth_inst, opacity_inst, inst_weights != 0);
  }
  return check_launch("composite_finish");
}

int64_t objnerf_ray_bias_floats(int64_t n_rays) { return n_rays < 0 ? -1 : n_rays * kRayBiasFloats; }

// One launch: the compact matrix of the hoisted weight columns was gathered when the weights were packed (it sits behind the
// aux block, objnerf_pack_weights) -- round 3 re-gathered it on every call, per ray set, pass and slab.  With a ray subset
// (objnerf_mlp_args.ray_index / n_active: render_rays_multi's culled object sets keep a fifth of the frame or less) only the
// listed rays are computed.
int objnerf_ray_bias(const objnerf_mlp_args* m, float* out, void* stream) {
  if (!m || !out || !m->blob || !m->aux || !m->rays || m->n_rays < 0) return set_error(-1, "ray_bias: bad arguments");
  if (m->do_object && !m->codes) return set_error(-1, "ray_bias: the object branch needs codes");
  if ((m->ray_index == nullptr) != (m->n_active == nullptr)) return set_error(-1, "ray_bias: ray_index and n_active go together");
  if (m->n_rays == 0) return 0;
  RayBiasArgs a{m->blob, m->aux, m->rays, m->codes, (long)m->code_stride, (long)m->n_rays, m->use_voxel, m->do_scene, m->do_object, out,
                m->ray_index, m->n_active};
  // a workgroup (two sets of four waves) per 64 rays, at most one per CU (98 KB of LDS: one is resident) -- it walks its pairs of
  // 32-ray groups with a stride
  const long pairs = (m->n_rays + 63) / 64;
  const dim3 grid(mlp_grid(pairs));
  if (m->do_scene && m->do_object) hipLaunchKernelGGL((ray_bias_kernel<true, true>), grid, dim3(512), 0, (hipStream_t)stream, a, m->aux + kAuxFloats);
  else if (m->do_scene) hipLaunchKernelGGL((ray_bias_kernel<true, false>), grid, dim3(512), 0, (hipStream_t)stream, a, m->aux + kAuxFloats);
  else if (m->do_object) hipLaunchKernelGGL((ray_bias_kernel<false, true>), grid, dim3(512), 0, (hipStream_t)stream, a, m->aux + kAuxFloats);
  else return 0;
  return check_launch("ray_bias");
}

int objnerf_generate_rays_rows(int H, int W, float focal, const float* h_c2w, float near, float far, const double* h_box,
                               double bbox_enlarge, int row0, int n_rows, int row_block, int block_stride, float* rays,
                               void* stream) {
  if (H < 1 || W < 1 || !(focal > 0.f) || !h_c2w) return set_error(-1, "generate_rays: bad arguments");
  if (n_rows < 0 || row0 < 0 || row_block < 1 || block_stride < 1) return set_error(-1, "generate_rays: bad row range");
  if (n_rows == 0) return 0;
  if (!rays) return set_error(-1, "generate_rays: null output");
  const int last = n_rows - 1;
  if (row0 + (last / row_block) * (long)row_block * block_stride + last % row_block >= H)
    return set_error(-1, "generate_rays: row range leaves the image");
  GenRaysArgs a;
  a.H = H; a.W = W; a.focal = focal; a.near = near; a.far = far;
  for (int i = 0; i < 12; ++i) a.c2w[i] = h_c2w[i];
  a.has_box = h_box != nullptr;
  for (int i = 0; i < OBJNERF_BOX_DOUBLES; ++i) a.box[i] = h_box ? h_box[i] : 0.0;
  a.enlarge = bbox_enlarge;
  a.row0 = row0; a.n_rows = n_rows; a.row_block = row_block; a.block_stride = block_stride;
  hipLaunchKernelGGL(generate_rays_kernel, dim3(blocks_for((long)n_rows * W, 256)), dim3(256), 0, (hipStream_t)stream, a, rays);
  return check_launch("generate_rays");
}

int objnerf_generate_rays(int H, int W, float focal, const float* h_c2w, float near, float far, const double* h_box,
                          double bbox_enlarge, float* rays, void* stream) {
  if (H < 1) return set_error(-1, "generate_rays: bad arguments");
  return objnerf_generate_rays_rows(H, W, focal, h_c2w, near, far, h_box, bbox_enlarge, 0, H, H, 1, rays, stream);
}

int objnerf_ray_directions(int H, int W, float focal, float* directions, void* stream) {
  if (H < 1 || W < 1 || !(focal > 0.f) || !directions) return set_error(-1, "ray_directions: bad arguments");
  hipLaunchKernelGGL(ray_directions_kernel, dim3(blocks_for((long)H * W, 256)), dim3(256), 0, (hipStream_t)stream, H, W, focal, directions);
  return check_launch("ray_directions");
}

int objnerf_get_rays(const float* directions, int64_t n, const float* c2w, int row_stride, float* rays_o, float* rays_d,
                     void* stream) {
  if (n < 0 || row_stride < 4) return set_error(-1, "get_rays: bad arguments");
  if (n == 0) return 0;
  if (!directions || !c2w || !rays_o || !rays_d) return set_error(-1, "get_rays: null pointer");
  hipLaunchKernelGGL(get_rays_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, directions, (long)n, c2w,
                     row_stride, rays_o, rays_d);
  return check_launch("get_rays");
}

int objnerf_ray_box_near_far(const float* rays_o, const float* rays_d, int64_t n, const double* h_box, double bbox_enlarge,
                             uint8_t* hit, float* near, float* far, void* stream) {
  if (n < 0 || !h_box) return set_error(-1, "ray_box_near_far: bad arguments");
  if (n == 0) return 0;
  if (!rays_o || !rays_d || !hit || !near || !far) return set_error(-1, "ray_box_near_far: null pointer");
  BoxArg b;
  for (int i = 0; i < OBJNERF_BOX_DOUBLES; ++i) b.box[i] = h_box[i];
  b.enlarge = bbox_enlarge;
  hipLaunchKernelGGL(ray_box_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, (long)n, b,
                     hit, near, far);
  return check_launch("ray_box_near_far");
}

int64_t objnerf_composite_multi_scratch_bytes(int K, int S) {
  if (K < 1 || S < 1) return -1;
  const size_t stage = (size_t)K * S * 7 * sizeof(float);
  return stage <= kMultiLdsMax ? 0 : (int64_t)(stage * kMultiGlobalGrid);
}

int objnerf_composite_multi(const objnerf_composite_multi_args* a, void* stream) {
  if (!a || a->K < 1 || a->K > kMaxSets || a->S < 1 || !a->h_z || !a->h_sigma || !a->h_rgb || !a->z_sorted ||
      !a->weights || !a->opacity || !a->rgb_map || !a->depth)
    return set_error(-1, "composite_multi: bad arguments (1 <= K <= 64)");
  const long M = (long)a->K * a->S;
  const size_t lds = (size_t)M * 7 * sizeof(float);
  const bool global = lds > kMultiLdsMax;
  if (global && !a->scratch)
    return set_error(-1, "composite_multi: K*S beyond the LDS staging limit needs `scratch` (objnerf_composite_multi_scratch_bytes)");
  if (a->noise_std != 0.f && !a->noise) return set_error(-1, "composite_multi: noise_std != 0 needs noise draws");
  if (a->n_rays == 0) return 0;
  MultiPtrs p;
  for (int k = 0; k < kMaxSets; ++k) {
    const bool on = k < a->K;
    p.z[k] = on ? a->h_z[k] : nullptr;
    p.sigma[k] = on ? a->h_sigma[k] : nullptr;
    p.rgb[k] = on ? a->h_rgb[k] : nullptr;
    p.own_w[k] = (on && a->h_own_weights) ? a->h_own_weights[k] : nullptr;
    if (on && (!p.z[k] || !p.sigma[k] || !p.rgb[k])) return set_error(-1, "composite_multi: null set pointer");
  }
  const float* noise = a->noise_std != 0.f ? a->noise : nullptr;
  if (global) {
    const unsigned grid = (unsigned)(a->n_rays < kMultiGlobalGrid ? a->n_rays : kMultiGlobalGrid);
    hipLaunchKernelGGL(composite_multi_kernel<true>, dim3(grid), dim3(64), 0, (hipStream_t)stream, p, (long)a->n_rays,
                       a->K, a->S, noise, a->noise_std, a->white_back, a->z_sorted, a->weights, a->obj_ids, a->opacity,
                       a->rgb_map, a->depth, a->h_own_weights ? 1 : 0, (float*)a->scratch);
    return check_launch("composite_multi(global staging)");
  }
  if (lds > 64 * 1024) {
    // beyond the default dynamic-LDS limit: raise this kernel's limit (a host-side attribute; once per process and device)
    static std::mutex mu;
    static bool raised[64] = {false};
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (dev >= 0 && dev < 64 && !raised[dev]) {
      if (hipFuncSetAttribute((const void*)composite_multi_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kMultiLdsMax) != hipSuccess)
        return set_error(-2, "composite_multi: could not raise the dynamic LDS limit");
      raised[dev] = true;
    }
  }
  unsigned grid = (unsigned)(a->n_rays < 65536 ? a->n_rays : 65536);
  hipLaunchKernelGGL(composite_multi_kernel<false>, dim3(grid), dim3(64), lds, (hipStream_t)stream, p, (long)a->n_rays,
                     a->K, a->S, noise, a->noise_std, a->white_back, a->z_sorted, a->weights, a->obj_ids, a->opacity,
                     a->rgb_map, a->depth, a->h_own_weights ? 1 : 0, nullptr);
  return check_launch("composite_multi");
}

int objnerf_pack_models(int use_voxel, const uint32_t* blob_idx, const uint32_t* aux_idx, int n_models,
                        const float* const* h_param_ptrs, float* const* h_blobs, float* const* h_auxs, void* stream) {
  if (!blob_idx || !aux_idx || !h_param_ptrs || !h_blobs || !h_auxs) return set_error(-1, "pack_models: bad arguments");
  if (n_models < 1 || n_models > kPackModels) return set_error(-1, "pack_models: 1 <= n_models <= 2");
  PackJobs jobs;
  for (int m = 0; m < n_models; ++m) {
    for (int i = 0; i < kNumParamPtrs; ++i) {
      if (!h_param_ptrs[m * kNumParamPtrs + i]) return set_error(-1, "pack_models: null parameter pointer");
      jobs.j[m].pp.p[i] = h_param_ptrs[m * kNumParamPtrs + i];
    }
    if (!h_blobs[m] || !h_auxs[m]) return set_error(-1, "pack_models: null output");
    jobs.j[m].blob = h_blobs[m];
    jobs.j[m].aux = h_auxs[m];
  }
  for (int m = n_models; m < kPackModels; ++m) jobs.j[m] = jobs.j[0];
  const long nb = objnerf_blob_floats(use_voxel), na = kAuxFloats;       // the tail of the aux block is the compact matrix
  if (nb % 4 != 0) return set_error(-3, "pack_models: stream length not a multiple of 4");
  const unsigned blocks = blocks_for(nb / 4, 256) + blocks_for(na, 256) + blocks_for(kRbMatFloats, 256);
  hipLaunchKernelGGL(pack_all_kernel, dim3(blocks, (unsigned)n_models), dim3(256), 0, (hipStream_t)stream, blob_idx, nb, aux_idx, na,
                     use_voxel, jobs);
  return check_launch("pack_models");
}

int objnerf_pack_weights(int use_voxel, const uint32_t* blob_idx, const uint32_t* aux_idx,
                         const float* const* h_param_ptrs, float* blob, float* aux, void* stream) {
  if (!blob || !aux) return set_error(-1, "pack_weights: bad arguments");
  float* blobs[1] = {blob};
  float* auxs[1] = {aux};
  return objnerf_pack_models(use_voxel, blob_idx, aux_idx, 1, h_param_ptrs, blobs, auxs, stream);
}

int objnerf_pack_weights_bwd(const uint32_t* blob_idx, const float* const* h_param_ptrs, float* blob, void* stream) {
  if (!blob_idx || !h_param_ptrs || !blob) return set_error(-1, "pack_weights_bwd: bad arguments");
  ParamPtrs pp;
  for (int i = 0; i < kNumParamPtrs; ++i) {
    if (!h_param_ptrs[i]) return set_error(-1, "pack_weights_bwd: null parameter pointer");
    pp.p[i] = h_param_ptrs[i];
  }
  const long nb = objnerf_bwd_blob_floats();
  hipLaunchKernelGGL(pack_kernel, dim3(blocks_for(nb, 256)), dim3(256), 0, (hipStream_t)stream, blob_idx, nb, pp, blob);
  return check_launch("pack_weights_bwd");
}

}  // extern "C"
