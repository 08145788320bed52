// device_math.h -- fp32 device helpers whose rounding behaviour is part of the parity contract.
// Compiled with -ffp-contract=off: `a * b + c` below is two roundings (like ATen's elementwise
// kernels), FMA only where fmaf() is written.
#pragma once
#include <hip/hip_runtime.h>

namespace objnerf {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// sin / cos with 3-term Cody-Waite reduction by pi/2 and the classic degree-7/8 minimax
// polynomials on [-pi/4, pi/4] (abs error ~1e-7 for |x| < 2^15, i.e. f32-roundoff class;
// the reference calls torch.sin/torch.cos, embedding_helper.py:69-74).
// Arguments beyond 2^15 (never produced by 2^k * coordinate; only reachable by a learned voxel
// feature > 1024) are answered branch-free by the hardware v_sin/v_cos on fract(x / 2pi), whose
// abs error there is ~1e-3 (documented limitation).
struct SinCos { float s, c; };

__device__ __forceinline__ SinCos psincos(float x) {
  const float n = rintf(x * 0.636619772367581343f);
  float r = fmaf(-n, 1.5703125f, x);
  r = fmaf(-n, 4.837512969970703125e-4f, r);
  r = fmaf(-n, 7.54978995489188216e-8f, r);
  const float z = r * r;
  float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(z, ps, -1.6666654611e-1f);
  const float s = fmaf(r * z, ps, r);
  float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(z, pc, 4.166664568298827e-2f);
  const float c = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
  const int q = (int)n;
  SinCos o;
  const float s1 = (q & 1) ? c : s;
  const float c1 = (q & 1) ? s : c;
  o.s = (q & 2) ? -s1 : s1;
  o.c = ((q + 1) & 2) ? -c1 : c1;
  const bool big = !(fabsf(x) < 32768.0f);
  const float rev = __builtin_amdgcn_fractf(x * 0.15915494309189535f);
  o.s = big ? __builtin_amdgcn_sinf(rev) : o.s;
  o.c = big ? __builtin_amdgcn_cosf(rev) : o.c;
  return o;
}
__device__ __forceinline__ float psin(float x) { return psincos(x).s; }
__device__ __forceinline__ float pcos(float x) { return psincos(x).c; }

// nn.LeakyReLU() default slope 0.01 (nerf_model.py:38)
__device__ __forceinline__ float leaky(float v) { return fmaxf(v, 0.01f * v); }

// torch.sigmoid
__device__ __forceinline__ float sigmoidf(float v) { return __fdiv_rn(1.0f, 1.0f + expf(-v)); }

}  // namespace objnerf
