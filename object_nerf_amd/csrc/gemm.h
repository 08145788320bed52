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
// other fp32 GEMM).  Operand tiles are staged global -> registers -> LDS with 16-byte accesses, double
// buffered (one barrier per k tile); a K-contiguous operand is stored [row][k] (rows padded to 36 floats) and
// read with one ds_read_b128 per 4 MFMA steps, a row-contiguous one [k][row] and read with ds_read_b32 --
// both conflict-free on the write and on the read side.  Split-K with fp32 atomics when the output has too
// few tiles to fill 256 CUs (wgrad: K = 10^5..10^6 points).
#pragma once
#include <hip/hip_runtime.h>
#include "device_math.h"

#include <utility>
#include <type_traits>

namespace objnerf {

template <int... Is, class F>
__device__ __forceinline__ void gemm_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}

enum GemmEpilogue { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_LEAKY = 2, EPI_BIAS_SIGMOID = 3, EPI_LEAKY_BWD = 4 };

struct GemmArgs {
  const float* A; long lda; int a_k_contig;   // A'[m][k] = a_k_contig ? A[m*lda + k] : A[k*lda + m]
  const float* B; long ldb; int b_k_contig;   // B'[k][n] = b_k_contig ? B[n*ldb + k] : B[k*ldb + n]
  float* C; long ldc;
  long M, N, K;
  int accumulate;        // C += (atomicAdd when split_k > 1)
  int epilogue;          // GemmEpilogue, applied to the complete sum (split_k must be 1)
  const float* bias;     // N floats; EPI_LEAKY_BWD: the layer's saved output, M x N with leading dimension ldc
  int split_k;           // >= 1
  float* rowsum;         // optional, row-contiguous A' only: rowsum[m] += sum_k A'[m][k] (a Linear layer's bias gradient
                         // when A' = dY^T), accumulated with atomics by the workgroups of the first column of tiles
  int bx0, nbx;          // column tiles [bx0, bx0 + nbx) of the problem are computed by this launch (set by gemm_launch)
  // Segmented contraction (split_k == 1 only): C = sum over segments s of A'_s [M x K_s] * B'_s [K_s x N], each segment
  // with its own operand pointers / leading dimensions and the same contiguity flags -- one launch and one epilogue
  // instead of nseg accumulating launches (the gradient w.r.t. an input that feeds several layers: train.hip).
  // nseg == 0: the single segment (A, lda, B, ldb, K) above.
  int nseg;
  const float* segA[4]; long seglda[4];
  const float* segB[4]; long segldb[4];
  long segK[4];
};

// operand rows of 271 / 439 / 527 / 27 floats (the embedding blocks and their weight columns) still move as dwordx4: f32x4u
// (device_math.h), a 16-byte load at any 4-byte-aligned address
constexpr int GBM = 128, GBN = 128, GBK = 32;
constexpr int GLDK = 36;      // K-contiguous operands sit in LDS as [row][k], rows padded to 36 floats against bank conflicts
constexpr int GLDR = 128;     // row-contiguous operands sit in LDS as [k][row]
constexpr int GTILE = GBM * GLDK;   // floats per staged operand tile (>= GBK * GLDR)
#ifndef OBJ_GEMM_DOUBLE_BUFFER
#define OBJ_GEMM_DOUBLE_BUFFER 0
#endif
#ifndef OBJ_GEMM_FRAG_PIPE
#define OBJ_GEMM_FRAG_PIPE 0   // full tiles: LDS fragment reads software-pipelined one 16-MFMA group ahead
#endif
#ifndef OBJ_GEMM_SPREAD_FETCH
// 1: full tiles: the eight 16-byte global loads of the next k tile are issued two at a time in front of the four 16-MFMA
// groups of the current one instead of as one burst (a burst queues at the CU's 64 B/clk vector-memory path and stalls the
// issuing waves -- what profiles/r03_wgrad_ablations.md measured for the weight-gradient loop)
#define OBJ_GEMM_SPREAD_FETCH 0
#endif
#ifndef OBJ_GEMM_SPREAD_FWD
// 1: the same for the forward shape only (both operands K-contiguous: Y = X W^T, the layer-wise MLP of generic.hip and of the
// OBJNERF_TRAIN_LAYERWISE path).  Measured on MI355X (profiles/r04_gemm_variants.txt): forward shapes 0.63 -> 0.67, 0.47 -> 0.54,
// 0.37 -> 0.45 of peak; the dX / dW shapes LOSE 4-8 % with it, so they keep the burst
#define OBJ_GEMM_SPREAD_FWD 1
#endif
#ifndef OBJ_GEMM_TAIL
// 1: a ragged last column tile with at most 3 of its four 32-column sub-tiles live is computed by a second launch of the
// TAIL instantiation: 4 x 1 wave layout, wave w owns row sub-tile w and the live column sub-tiles, so that tile costs
// ncol / 4 of a full one instead of all of it (271 columns = 2 tiles + 15 columns: 9 units of work instead of 12).
#define OBJ_GEMM_TAIL 1
#endif
#ifndef OBJ_GEMM_XCD
// 1: the output tiles that stream the same operand panel get consecutive slots on ONE XCD (shared L2).  Measured
// (tools/gemm_bench.py, round 2, libraries interleaved after a warm-up): neutral -- forward 0.66 vs 0.65 of peak, dgrad
// 0.69 vs 0.70, wgrad 256 x 256 0.78 vs 0.76: these products are not HBM-bound (a panel re-read by a workgroup on
// another XCD comes from the MALL), so the plain map stays.
#define OBJ_GEMM_XCD 0
#endif
// 1-D launch grid of gemm_kernel for a problem (host side, train.hip)
inline unsigned gemm_grid(long M, unsigned nx, int split_k) {
  const unsigned ny = (unsigned)((M + 127) / 128);
#if OBJ_GEMM_XCD
  const unsigned groups = split_k > 1 ? (unsigned)split_k : ny, per_group = split_k > 1 ? nx * ny : nx;
  return 8u * per_group * ((groups + 7) / 8);
#else
  return nx * ny * (unsigned)split_k;
#endif
}

// One operand of a workgroup: 128 "rows" (m for A', n for B') x 32 k per tile.  Global -> registers (16-byte loads
// whenever the operand allows it) -> LDS in the layout its MFMA reads want; the fetch of tile t+1 is issued before
// the MFMAs of tile t and written to the other LDS buffer after them, so its latency hides under 64 MFMAs per wave.
// Contraction index of MFMA step (s4, s) in lane half h is k = 16 h + 4 s4 + s for both operands (any bijection
// is a valid summation order), which makes a K-contiguous row a plain copy: no shuffle on the way to LDS.
template <bool K_CONTIG>
struct GemmOperand {
  const float* p[4];     // this thread's four 16-byte pieces of the current tile
  long step;             // pointer advance per k tile
  long rows_left;        // row-contiguous: rows - (first of this thread's 4 rows); K-contiguous: unused
  bool ok[4];            // K-contiguous: piece's row is inside the operand
  bool fast;             // uniform: every full k tile can be fetched with unconditional 16-byte loads
  bool over;             // uniform, row-contiguous only: the operand has fewer than 128 rows from row0 on, but a 128-row read of
                         // any k line that has GBK more lines after it stays inside the matrix (it runs on into the next lines):
                         // such k tiles are fetched whole as well -- the surplus rows only meet output rows / columns that are
                         // never stored (measured on the 104-column object-voxel products, profiles/r04_train_ab.txt)
  bool aligned;          // ... whose addresses are 16-byte aligned (else the unaligned-vector form of the same load)
  f32x4 v[4];

  __device__ __forceinline__ void init(const float* src, long ld, long row0, long rows, long kbeg, int tid) {
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    if (K_CONTIG) {
      // thread -> rows tid/8 + 32 i, k = 4 (tid % 8) .. + 3.  Rows past the end are clamped, not zeroed: an output
      // row depends only on its own operand row, and rows past the end are never stored.
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long gr = row0 + (tid >> 3) + 32 * i;
        ok[i] = gr < rows;
        p[i] = src + (ok[i] ? gr : rows - 1) * ld + kbeg + 4 * (tid & 7);
      }
      step = GBK;
      rows_left = 0;
      fast = true;
      over = false;
      aligned = vec;
    } else {
      // thread -> k = tid/32 + 8 i, rows 4 (tid % 32) .. + 3
      const long r = row0 + 4 * (tid & 31);
      rows_left = rows - r;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ok[i] = true;
        p[i] = src + (kbeg + (tid >> 5) + 8 * i) * ld + (rows_left > 0 ? r : 0);
      }
      step = GBK * ld;
      fast = row0 + GBM <= rows;
      over = !fast && row0 < rows && row0 + GBM <= (long)(GBK + 1) * ld;
      aligned = vec;
    }
  }
  // k0: first k of the tile being fetched.  The contraction range must be zero-filled in BOTH operands
  // (0 x garbage could be NaN), which only the last tile of a split needs.
  __device__ __forceinline__ void fetch(long k0, long kend, int tid) {
    if ((fast && k0 + GBK <= kend) || (over && k0 + 2 * GBK <= kend)) {            // uniform branches
      if (aligned) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = gload4(p[i]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = gload4u(p[i]);
      }
    } else if (K_CONTIG) {
      const long gk = k0 + 4 * (tid & 7);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][j] = (gk + j < kend) ? gload(p[i] + j) : 0.f;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool kin = k0 + (tid >> 5) + 8 * i < kend;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][j] = (kin && j < rows_left) ? gload(p[i] + j) : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] += step;
  }
  // one piece of the next tile (unconditional 16-byte load: only when can_spread()); advance() after all four
  __device__ __forceinline__ bool can_spread(long k0, long kend) const { return fast && k0 + GBK <= kend; }
  template <int I>
  __device__ __forceinline__ void fetch_one() { v[I] = aligned ? gload4(p[I]) : gload4u(p[I]); }
  __device__ __forceinline__ void advance() {
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] += step;
  }
  __device__ __forceinline__ void put(float* lds, int tid) const {
    if (K_CONTIG) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *(f32x4*)&lds[((tid >> 3) + 32 * i) * GLDK + 4 * (tid & 7)] = v[i];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) *(f32x4*)&lds[((tid >> 5) + 8 * i) * GLDR + 4 * (tid & 31)] = v[i];
    }
  }
  // the four MFMA operands (steps s = 0..3 of group s4) of tile row `row` for lane half `half`
  static __device__ __forceinline__ f32x4 frag(const float* lds, int row, int half, int s4) {
    if (K_CONTIG) return *(const f32x4*)&lds[row * GLDK + 16 * half + 4 * s4];
    f32x4 t;
#pragma unroll
    for (int s = 0; s < 4; ++s) t[s] = lds[(16 * half + 4 * s4 + s) * GLDR + row];
    return t;
  }
};

#ifndef OBJ_GEMM_WAVES
#define OBJ_GEMM_WAVES 0     // > 0: ask the register allocator for this many waves per SIMD (tuning switch)
#endif
#if OBJ_GEMM_WAVES
#define OBJ_GEMM_OCC __attribute__((amdgpu_waves_per_eu(OBJ_GEMM_WAVES, OBJ_GEMM_WAVES)))
#else
#define OBJ_GEMM_OCC
#endif
template <bool A_KC, bool B_KC, bool TAIL = false>
__global__ void __launch_bounds__(256) OBJ_GEMM_OCC gemm_kernel(const GemmArgs g) {
  constexpr int NBUF = OBJ_GEMM_DOUBLE_BUFFER ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float lds[NBUF * 2 * GTILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // Workgroup -> (tile, k slice).  OBJ_GEMM_XCD (off, see above): workgroups are dealt round-robin to the 8 XCDs
  // (linear id % 8); the output tiles that stream the SAME operand panel -- all tiles of one k slice (split-K: wgrad),
  // or the tiles of one row panel (split_k == 1) -- get consecutive slots on ONE XCD.
  const unsigned nx = (unsigned)g.nbx, ny = (unsigned)((g.M + GBM - 1) / GBM);
#if OBJ_GEMM_XCD
  const unsigned groups = g.split_k > 1 ? (unsigned)g.split_k : ny, per_group = g.split_k > 1 ? nx * ny : nx;
  const unsigned slot = blockIdx.x >> 3;
  const unsigned grp = (slot / per_group) * 8 + (blockIdx.x & 7), t = slot % per_group;
  if (grp >= groups) return;                       // padding of the last round of 8 groups
  const unsigned bz = g.split_k > 1 ? grp : 0, bx = t % nx, by = g.split_k > 1 ? t / nx : grp;
#else
  const unsigned bx = blockIdx.x % nx, by = (blockIdx.x / nx) % ny, bz = blockIdx.x / (nx * ny);
#endif
  const long m0 = (long)by * GBM, n0 = (long)(g.bx0 + bx) * GBN;
  // TAIL: live 32-column sub-tiles of this (last, ragged) column tile
  const int ncol = TAIL ? (int)((g.N - n0 + 31) / 32) : 4;

  // full tile: 2 x 2 waves, each 2 x 2 MFMA tiles (acc[2 i + j]); TAIL: 4 x 1 waves, acc[j] = column sub-tile j
  f32x16 acc[TAIL ? 3 : 4];
#pragma unroll
  for (int i = 0; i < (TAIL ? 3 : 4); ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  float rsum = 0.f;
  const bool want_rowsum = !A_KC && g.rowsum != nullptr && g.bx0 + bx == 0;     // uniform
  const int nseg = g.nseg > 0 ? g.nseg : 1;
  for (int sg = 0; sg < nseg; ++sg) {
  const float* segA = g.nseg > 0 ? g.segA[sg] : g.A;
  const float* segB = g.nseg > 0 ? g.segB[sg] : g.B;
  const long seglda = g.nseg > 0 ? g.seglda[sg] : g.lda, segldb = g.nseg > 0 ? g.segldb[sg] : g.ldb;
  const long segK = g.nseg > 0 ? g.segK[sg] : g.K;
  // split-K slice of this workgroup (a segmented contraction is never split)
  const long kchunk = ((segK + g.split_k - 1) / g.split_k + GBK - 1) / GBK * GBK;
  const long kbeg = (long)bz * kchunk;
  const long kend = kbeg + kchunk < segK ? kbeg + kchunk : segK;
  if (kbeg < kend) {          // uniform per workgroup
    GemmOperand<A_KC> opa;
    GemmOperand<B_KC> opb;    // B'[k][n]: "row" of the staged tile = n
    opa.init(segA, seglda, m0, g.M, kbeg, tid);
    opb.init(segB, segldb, n0, g.N, kbeg, tid);
    opa.fetch(kbeg, kend, tid);
    opb.fetch(kbeg, kend, tid);
    const int half = lane >> 5, rl = lane & 31;
    int buf = 0;
    if (NBUF == 2) {
      opa.put(lds, tid);
      opb.put(lds + GTILE, tid);
      __syncthreads();
    }
    for (long k0 = kbeg; k0 < kend; k0 += GBK) {
      const bool more = k0 + GBK < kend;
      float* As = lds + buf * 2 * GTILE;
      float* Bs = As + GTILE;
      if (NBUF == 1) {
        __syncthreads();                     // every wave is done reading the previous tile
        opa.put(As, tid);
        opb.put(Bs, tid);
        __syncthreads();
      }
      constexpr bool kSpread = OBJ_GEMM_SPREAD_FETCH || (OBJ_GEMM_SPREAD_FWD && A_KC && B_KC);
      const bool spread = kSpread && !TAIL && more && opa.can_spread(k0 + GBK, kend) && opb.can_spread(k0 + GBK, kend);   // uniform
      if (more && !spread) {                 // next tile's global loads fly under this tile's MFMAs
        opa.fetch(k0 + GBK, kend, tid);
        opb.fetch(k0 + GBK, kend, tid);
      }
      if (want_rowsum) {                     // A' tile is [k][row]: thread -> row tid % 128, 16 of the 32 k
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) rsum += As[((tid >> 7) * 16 + kk) * GLDR + (tid & 127)];
      }
      if constexpr (TAIL) {
        if (m0 + wave * 32 < g.M) {          // wave-uniform: this wave's row sub-tile exists
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            const f32x4 a = GemmOperand<A_KC>::frag(As, wave * 32 + rl, half, s4);
            f32x4 b[3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
              if (j < ncol) b[j] = GemmOperand<B_KC>::frag(Bs, j * 32 + rl, half, s4);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int j = 0; j < 3; ++j)
                if (j < ncol) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[j][s], acc[j], 0, 0, 0);
          }
        }
      } else {
#if OBJ_GEMM_FRAG_PIPE
        // fragments of group s4 + 1 are read from LDS before the 16 MFMAs of group s4 issue
        f32x4 fa[2][2], fb[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[0][i] = GemmOperand<A_KC>::frag(As, wm * 64 + i * 32 + rl, half, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[0][j] = GemmOperand<B_KC>::frag(Bs, wn * 64 + j * 32 + rl, half, 0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          if (s4 < 3) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[(s4 + 1) & 1][i] = GemmOperand<A_KC>::frag(As, wm * 64 + i * 32 + rl, half, s4 + 1);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[(s4 + 1) & 1][j] = GemmOperand<B_KC>::frag(Bs, wn * 64 + j * 32 + rl, half, s4 + 1);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                acc[2 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s4 & 1][i][s], fb[s4 & 1][j][s], acc[2 * i + j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
#else
        gemm_static_for_impl([&](auto S4) __attribute__((always_inline)) {       // 4 MFMA steps per fragment read
          constexpr int s4 = decltype(S4)::value;
          f32x4 a[2], b[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) a[i] = GemmOperand<A_KC>::frag(As, wm * 64 + i * 32 + rl, half, s4);
#pragma unroll
          for (int j = 0; j < 2; ++j) b[j] = GemmOperand<B_KC>::frag(Bs, wn * 64 + j * 32 + rl, half, s4);
          if constexpr (kSpread) {
            if (spread) {                      // this group's share of the next tile's loads
              opa.template fetch_one<s4>();
              opb.template fetch_one<s4>();
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                acc[2 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[2 * i + j], 0, 0, 0);
          if constexpr (kSpread) __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, 4>{});
        if (spread) { opa.advance(); opb.advance(); }
#endif
      }
      if (NBUF == 2) {
        if (more) {                          // the other buffer was last read before the previous barrier
          opa.put(lds + (buf ^ 1) * 2 * GTILE, tid);
          opb.put(lds + (buf ^ 1) * 2 * GTILE + GTILE, tid);
        }
        __syncthreads();
        buf ^= 1;
      }
    }
  }
  if (nseg > 1) __syncthreads();     // the next segment's first tile overwrites the LDS buffers
  }
  if (want_rowsum && m0 + (tid & 127) < g.M) atomicAdd(g.rowsum + m0 + (tid & 127), rsum);
  // D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int col = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int t = 0; t < (TAIL ? 3 : 4); ++t) {
    {
      const int i = t >> 1, j = t & 1;
      if (TAIL && t >= ncol) continue;
      const long n = TAIL ? n0 + t * 32 + col : n0 + wn * 64 + j * 32 + col;
      if (n >= g.N) continue;
      const bool lbwd = g.epilogue == EPI_LEAKY_BWD;
      const float bv = (g.epilogue != EPI_NONE && !lbwd && g.bias) ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = (TAIL ? m0 + wave * 32 : m0 + wm * 64 + i * 32) + (r & 3) + 8 * (r >> 2) + rbase;
        if (m >= g.M) continue;
        float* c = g.C + m * g.ldc + n;
        float v = acc[t][r];
        if (g.split_k > 1) { atomicAdd(c, v); continue; }
        if (g.accumulate) v += *c;
        if (lbwd) {                  // leaky_relu backward on sign(output) = sign(input)
          v = g.bias[m * g.ldc + n] > 0.f ? v : 0.01f * v;
        } else if (g.epilogue != EPI_NONE) {
          v += bv;
          if (g.epilogue == EPI_BIAS_LEAKY) v = leaky(v);
          else if (g.epilogue == EPI_BIAS_SIGMOID) v = sigmoidf(v);
        }
        *c = v;
      }
    }
  }
}

// host launcher (train.hip)
int gemm_launch(const GemmArgs& g, hipStream_t s);

}  // namespace objnerf
