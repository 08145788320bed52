// device_math.h -- fp32 device helpers whose rounding behaviour is part of the parity contract.
// Compiled with -ffp-contract=off: `a * b + c` below is two roundings (like ATen's elementwise
// kernels), FMA only where fmaf() is written.
#pragma once
#include <hip/hip_runtime.h>

namespace objnerf {

using f32x4 = __attribute__((ext_vector_type(4))) float;
// four consecutive floats at ANY 4-byte-aligned address as one 16-byte access (gfx950 global memory needs dword alignment only)
typedef f32x4 f32x4u __attribute__((aligned(4)));
using f32x16 = __attribute__((ext_vector_type(16))) float;

// Accesses through a pointer the compiler cannot prove global (one read out of a device-memory argument list, wgrad.h)
// compile to flat_* instructions, which count on lgkmcnt as well as vmcnt: every later LDS wait then also waits for the
// prefetch in flight.  These say "global memory" explicitly (global_load / global_store, vmcnt only).
#define OBJ_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ f32x4 gload4(const float* p) { return *(const OBJ_GLOBAL f32x4*)p; }
__device__ __forceinline__ f32x4 gload4u(const float* p) { return *(const OBJ_GLOBAL f32x4u*)p; }
__device__ __forceinline__ float gload(const float* p) { return *(const OBJ_GLOBAL float*)p; }
__device__ __forceinline__ void gstore(float* p, float v) { *(OBJ_GLOBAL float*)p = v; }
// "this value is the same in every lane": moves it to scalar registers, so that what is computed from it (loop counters,
// base addresses) runs on the SALU and global addresses take the scalar-base form
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ long uni(long x) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long)x >> 32));
  return (long)(((unsigned long)hi << 32) | lo);
}
template <class T> __device__ __forceinline__ T* uni(T* p) { return reinterpret_cast<T*>(uni(reinterpret_cast<long>(p))); }

// sin / cos with 3-term Cody-Waite reduction by pi/2 and the classic degree-7/8 minimax
// polynomials on [-pi/4, pi/4] (abs error ~1e-7 for |x| < 2^15, i.e. f32-roundoff class;
// the reference calls torch.sin/torch.cos, embedding_helper.py:69-74).
// The reduction is exact-product safe while n = x * 2/pi < 2^16, i.e. |x| < 65536 (2^9 * coordinate
// stays below that for |coordinate| < 128 normalised units; a learned voxel feature would have to
// exceed 2048).  Beyond it:
//   EXACT_BIG = true  (stand-alone embedding kernels): branch to OCML sinf/cosf (Payne-Hanek);
//   EXACT_BIG = false (fused MLP kernel): no special handling -- fp32 MFMA and VALU share the
//     SIMD's ALUs (PMC: MFMA-busy + VALU-active + waits = 100 %), so every VALU instruction here
//     costs MFMA time.  Accuracy then degrades progressively with |x| / 65536 (output stays
//     bounded); unreachable by 2^k * coordinate inside a scene (documented limitation).
struct SinCos { float s, c; };
constexpr float kSinCosBig = 65536.0f;

template <bool EXACT_BIG = false>
__device__ __forceinline__ SinCos psincos(float x) {
  if constexpr (EXACT_BIG) {
    if (__builtin_expect(!(fabsf(x) < kSinCosBig), 0)) {
      SinCos o;
      o.s = sinf(x);
      o.c = cosf(x);
      return o;
    }
  }
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
  return o;
}

// nn.LeakyReLU() default slope 0.01 (nerf_model.py:38)
__device__ __forceinline__ float leaky(float v) { return fmaxf(v, 0.01f * v); }

// torch.sigmoid
__device__ __forceinline__ float sigmoidf(float v) { return __fdiv_rn(1.0f, 1.0f + expf(-v)); }

}  // namespace objnerf
