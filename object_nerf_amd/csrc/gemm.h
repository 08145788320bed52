// gemm.h -- fp32 MFMA GEMM used by the training path.
//
//   C[M x N] = alpha * A'[M x K] * B'[K x N] (+ C)   with an optional fused epilogue
//
// A' and B' are described by (pointer, leading dimension, which index is contiguous), which covers the
// three products of a Linear layer on row-major activations without materialising any transpose:
//   forward  Y  = X  * W^T :  A' = X  (K contiguous), B' = W   (K contiguous: B'[k][n] = W[n][k])
//   dgrad    dX = dY * W   :  A' = dY (K contiguous), B' = W   (N contiguous: B'[k][n] = W[k][n])
//   wgrad    dW = dY^T * X :  A' = dY (M contiguous: A'[m][k] = dY[k][m]), B' = X (N contiguous), K = #points
//
// 128 x 128 x 32 tile per 256-thread workgroup, 2 x 2 waves each owning 2 x 2 tiles of
// v_mfma_f32_32x32x2_f32 (exact fp32 = an fmaf chain, so results are fp32-roundoff identical to any
// other fp32 GEMM).  Operand tiles go through LDS with the k index stored as (k & 1) * 16 + (k >> 1), so
// that the two lane halves of an MFMA (which contract k = 2s and 2s + 1) each read 16 contiguous floats
// (4 x ds_read_b128 per 16 MFMA steps); rows are padded to 36 floats against bank conflicts.  Split-K with
// fp32 atomics when the output has too few tiles to fill 256 CUs (wgrad: K = 10^5..10^6 points).
// This is a correctness-first kernel for the non-headline training path, not a tuned SGEMM.
#pragma once
#include <hip/hip_runtime.h>
#include "device_math.h"

namespace objnerf {

enum GemmEpilogue { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_LEAKY = 2, EPI_BIAS_SIGMOID = 3 };

struct GemmArgs {
  const float* A; long lda; int a_k_contig;   // A'[m][k] = a_k_contig ? A[m*lda + k] : A[k*lda + m]
  const float* B; long ldb; int b_k_contig;   // B'[k][n] = b_k_contig ? B[n*ldb + k] : B[k*ldb + n]
  float* C; long ldc;
  long M, N, K;
  int accumulate;        // C += (atomicAdd when split_k > 1)
  int epilogue;          // GemmEpilogue, applied to the complete sum (split_k must be 1)
  const float* bias;     // N floats
  int split_k;           // >= 1
};

constexpr int GBM = 128, GBN = 128, GBK = 32, GLD = 36;

// Staging is split in two (issue-early / write-late): the 16 floats a thread contributes to the next A' and B'
// tiles are fetched into registers BEFORE the MFMAs of the current tile and written to LDS after them, so the
// global-load latency hides under 32 MFMAs per wave.
template <bool K_CONTIG>
__device__ __forceinline__ void gemm_fetch(float (&v)[16], const float* src, long ld, long row0, long k0, long rows,
                                           long kend, int tid) {
  if (K_CONTIG) {
    // thread -> (row = tid/8 + 32*i, 4 consecutive k = (tid%8)*4)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (tid >> 3) + 32 * i, kk = (tid & 7) * 4;
      const long gr = row0 + r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long gk = k0 + kk + j;
        v[i * 4 + j] = (gr < rows && gk < kend) ? src[gr * ld + gk] : 0.f;
      }
    }
  } else {
    // row index contiguous in memory: thread -> (k = tid/32 + 8*i, 4 consecutive rows = (tid%32)*4)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = (tid >> 5) + 8 * i, rr = (tid & 31) * 4;
      const long gk = k0 + k;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long gr = row0 + rr + j;
        v[i * 4 + j] = (gr < rows && gk < kend) ? src[gk * ld + gr] : 0.f;
      }
    }
  }
}
template <bool K_CONTIG>
__device__ __forceinline__ void gemm_put(float* lds, const float (&v)[16], int tid) {
  // lds[r][perm(k)], perm(k) = (k & 1) * 16 + (k >> 1)
  if (K_CONTIG) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (tid >> 3) + 32 * i, kk = (tid & 7) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = kk + j;
        lds[r * GLD + (k & 1) * 16 + (k >> 1)] = v[i * 4 + j];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = (tid >> 5) + 8 * i, rr = (tid & 31) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) lds[(rr + j) * GLD + (k & 1) * 16 + (k >> 1)] = v[i * 4 + j];
    }
  }
}

template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[GBM * GLD];
  __shared__ __attribute__((aligned(16))) float Bs[GBN * GLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long m0 = (long)blockIdx.y * GBM, n0 = (long)blockIdx.x * GBN;
  // split-K slice of this workgroup
  const long kchunk = ((g.K + g.split_k - 1) / g.split_k + GBK - 1) / GBK * GBK;
  const long kbeg = (long)blockIdx.z * kchunk;
  const long kend = kbeg + kchunk < g.K ? kbeg + kchunk : g.K;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float va[16], vb[16];
  if (kbeg < kend) {
    gemm_fetch<A_KC>(va, g.A, g.lda, m0, kbeg, g.M, kend, tid);
    gemm_fetch<B_KC>(vb, g.B, g.ldb, n0, kbeg, g.N, kend, tid);     // B'[k][n]: "row" of the staged tile = n
  }
  for (long k0 = kbeg; k0 < kend; k0 += GBK) {
    __syncthreads();                       // every wave is done reading the previous tile
    gemm_put<A_KC>(As, va, tid);
    gemm_put<B_KC>(Bs, vb, tid);
    __syncthreads();
    if (k0 + GBK < kend) {                 // next tile's global loads fly under this tile's MFMAs
      gemm_fetch<A_KC>(va, g.A, g.lda, m0, k0 + GBK, g.M, kend, tid);
      gemm_fetch<B_KC>(vb, g.B, g.ldb, n0, k0 + GBK, g.N, kend, tid);
    }
    const int half = lane >> 5, rl = lane & 31;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {       // 4 MFMA steps (k pairs) per 16-byte read
      f32x4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *(const f32x4*)&As[(wm * 64 + i * 32 + rl) * GLD + half * 16 + s4 * 4];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *(const f32x4*)&Bs[(wn * 64 + j * 32 + rl) * GLD + half * 16 + s4 * 4];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    }
  }
  // D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int col = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long n = n0 + wn * 64 + j * 32 + col;
      if (n >= g.N) continue;
      const float bv = (g.epilogue != EPI_NONE && g.bias) ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rbase;
        if (m >= g.M) continue;
        float* c = g.C + m * g.ldc + n;
        float v = acc[i][j][r];
        if (g.split_k > 1) { atomicAdd(c, v); continue; }
        if (g.accumulate) v += *c;
        if (g.epilogue != EPI_NONE) {
          v += bv;
          if (g.epilogue == EPI_BIAS_LEAKY) v = leaky(v);
          else if (g.epilogue == EPI_BIAS_SIGMOID) v = sigmoidf(v);
        }
        *c = v;
      }
    }
}

// host launcher (train.hip)
int gemm_launch(const GemmArgs& g, hipStream_t s);

}  // namespace objnerf
