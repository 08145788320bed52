// Micro-probe (not part of the library): how fast is "fp32 via 3-way bf16 split" on gfx950's bf16 matrix pipe, per wave
// at 1 wave/SIMD, next to the fp32 MFMA this repo's MLP kernel uses?  One k16-step of a 32-point x 256-output layer:
//   fp32 : 8 out tiles x 8 v_mfma_f32_32x32x2_f32                               = 64 MFMAs (64 cycles each)
//   split: 8 values/lane -> (hi, mid, lo) bf16 (truncation splits are exact: 8+8+8 bits), 8 tiles x 6 v_mfma_f32_32x32x16_bf16
// A operands come from LDS like in the kernel.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/bf16x3_probe.hip -o /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 1) k_fp32(float* out, int iters) {
  __shared__ f32x4 lds[8 * 2 * 64];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8 * 2 * 64; i += 256) lds[i] = f32x4{1e-3f * i, 2e-3f, 3e-3f, 4e-3f};
  __syncthreads();
  f32x16 acc[8];
  for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  float b[8];
  for (int j = 0; j < 8; ++j) b[j] = 1e-3f * (lane + j);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f32x4 a[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) a[m] = lds[(m * 2 + g) * 64 + lane];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][j], b[g * 4 + j], acc[m], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = b[j] * 1.0001f + acc[j][0] * 1e-20f;      // next "activations" (VALU, data dependent)
  }
  float s = 0.f;
  for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__device__ __forceinline__ unsigned hi16(float x) { return __float_as_uint(x) & 0xffff0000u; }

template <int VALU_EXTRA>
__global__ void __launch_bounds__(256, 1) k_split(float* out, int iters) {
  __shared__ u32x4 lds[8 * 3 * 64];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8 * 3 * 64; i += 256) lds[i] = u32x4{0x3c003c00u + i, 0x3c003c00u, 0x3b003b00u, 0x3a003a00u};
  __syncthreads();
  f32x16 acc[8];
  for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  float b[8];
  for (int j = 0; j < 8; ++j) b[j] = 1e-3f * (lane + j);
  for (int it = 0; it < iters; ++it) {
    // split 8 fp32 values into three bf16 planes (truncation: exact), packed two per register
    u32x4 ph, pm, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x0 = b[2 * j], x1 = b[2 * j + 1];
      const unsigned h0 = hi16(x0), h1 = hi16(x1);
      const float r0 = x0 - __uint_as_float(h0), r1 = x1 - __uint_as_float(h1);
      const unsigned m0 = hi16(r0), m1 = hi16(r1);
      const float s0 = r0 - __uint_as_float(m0), s1 = r1 - __uint_as_float(m1);
      ph[j] = (h0 >> 16) | h1;
      pm[j] = (m0 >> 16) | m1;
      pl[j] = (hi16(s0) >> 16) | hi16(s1);
    }
    const bf16x8 bh = __builtin_bit_cast(bf16x8, ph), bm = __builtin_bit_cast(bf16x8, pm), bl = __builtin_bit_cast(bf16x8, pl);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const bf16x8 ah = __builtin_bit_cast(bf16x8, lds[(m * 3 + 0) * 64 + lane]);
      const bf16x8 am = __builtin_bit_cast(bf16x8, lds[(m * 3 + 1) * 64 + lane]);
      const bf16x8 al = __builtin_bit_cast(bf16x8, lds[(m * 3 + 2) * 64 + lane]);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[m], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      b[j] = b[j] * 1.0001f + acc[j][0] * 1e-20f;
#pragma unroll
      for (int e = 0; e < VALU_EXTRA; ++e) b[j] = b[j] * 0.99999f + 1e-9f;       // stand-in for sin/cos / epilogue VALU work
    }
  }
  float s = 0.f;
  for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class K>
static float time_it(K launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  const int iters = 200000;
  // per iteration and wave: one k16-step of a 256-output layer for 32 points = 2 * 32 * 256 * 16 FLOP
  const double flop = 256.0 * 4 * iters * 2.0 * 32 * 256 * 16;
  float t = time_it([&] { hipLaunchKernelGGL(k_fp32, dim3(256), dim3(256), 0, 0, out, iters); });
  printf("fp32 MFMA 32x32x2      : %8.2f ms  %7.1f TFLOP/s (fp32-equivalent)\n", t, flop / t / 1e9);
  t = time_it([&] { hipLaunchKernelGGL((k_split<0>), dim3(256), dim3(256), 0, 0, out, iters); });
  printf("bf16x3, 6 products     : %8.2f ms  %7.1f TFLOP/s (fp32-equivalent)\n", t, flop / t / 1e9);
  t = time_it([&] { hipLaunchKernelGGL((k_split<4>), dim3(256), dim3(256), 0, 0, out, iters); });
  printf("bf16x3 + 64 extra VALU : %8.2f ms  %7.1f TFLOP/s (fp32-equivalent)\n", t, flop / t / 1e9);
  t = time_it([&] { hipLaunchKernelGGL((k_split<16>), dim3(256), dim3(256), 0, 0, out, iters); });
  printf("bf16x3 + 256 extra VALU: %8.2f ms  %7.1f TFLOP/s (fp32-equivalent)\n", t, flop / t / 1e9);
  return 0;
}
