// wgrad.hip -- grouped, stream-K, deterministic weight gradients of one backward pass (wgrad.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "wgrad.h"
#include "host_api.h"

namespace objnerf {

// ---- ragged tiles (1..3 live 32-column sub-tiles): 4 x 1 waves, gemm.h's operand staging, single LDS buffer ----------------
// (Round 6 tried two register sets per operand -- the tile after next in flight under two tiles' MFMAs: ~190 VGPRs, i.e. two waves
// per SIMD instead of three, and the step measured 18.19 vs 18.07 ms without it (three waves with the spills: 18.66),
// profiles/r06_train_ab_fixup.txt -- the occupancy it costs hides as much latency as the deeper prefetch.)
__device__ __forceinline__ void wgrad_tail_piece(const WgProduct& pr, const WgTile& tl, long kbeg, long kend,
                                                 float* slot, float* lds, int tid, long P) {
  const long nseg16 = (P + 15) >> 4;
  const int lane = tid & 63, wave = tid >> 6;
  const long m0 = (long)tl.by * GBM, n0 = (long)tl.bx * GBN;
  const int ncol = tl.ncol;
  f32x16 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float rsum = 0.f;
  const bool want_rowsum = pr.rowsum != nullptr && tl.bx == 0;
  GemmOperand<false> opa, opb;
  opa.init(pr.A, pr.lda, m0, pr.M, kbeg, tid);
  opb.init(pr.B, pr.ldb, n0, pr.N, kbeg, tid);
  opa.fetch(kbeg, kend, tid);
  opb.fetch(kbeg, kend, tid);
  const int half = lane >> 5, rl = lane & 31;
  float* As = lds;
  float* Bs = lds + GTILE;
  for (long k0 = kbeg; k0 < kend; k0 += GBK) {
    const bool more = k0 + GBK < kend;
    __syncthreads();                     // every wave is done reading the previous tile
    opa.put(As, tid);
    opb.put(Bs, tid);
    __syncthreads();
    if (more) {                          // next tile's global loads fly under this tile's MFMAs
      opa.fetch(k0 + GBK, kend, tid);
      opb.fetch(k0 + GBK, kend, tid);
    }
    if (want_rowsum) {                   // A' tile is [k][row]: thread -> row tid % 128, 16 of the 32 k
      float s16 = 0.f;                   // (the 16-point column sums of WgProduct::segsum; rsum keeps its own chain of additions)
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) { const float x = As[((tid >> 7) * 16 + kk) * GLDR + (tid & 127)]; rsum += x; s16 += x; }
      if (pr.segsum) {
        const long seg = (k0 >> 4) + (tid >> 7);
        if (seg < nseg16 && m0 + (tid & 127) < pr.M) gstore(pr.segsum + seg * pr.seg_ld + m0 + (tid & 127), s16);
      }
    }
    if (m0 + wave * 32 < pr.M) {         // wave-uniform: this wave's row sub-tile exists
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const f32x4 a = GemmOperand<false>::frag(As, wave * 32 + rl, half, s4);
        f32x4 b[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (j < ncol) b[j] = GemmOperand<false>::frag(Bs, j * 32 + rl, half, s4);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int j = 0; j < 3; ++j)
            if (j < ncol) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[j][s], acc[j], 0, 0, 0);
      }
    }
  }
  __syncthreads();                       // the row-sum exchange below reuses the LDS buffers
  // row sums: the two 16-k halves of a row live in threads tid and tid + 128: combine through LDS (fixed order)
  if (want_rowsum) {
    if (tid >= 128) lds[tid - 128] = rsum;
    __syncthreads();
    if (tid < 128) gstore(slot + 128 * 128 + tid, rsum + lds[tid]);
  }
  // D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); the partial tile goes to the unit's slot
  // as a dense 128 x 128 block (rows / columns past the product's extent are never read back)
  const int col = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    if (t >= ncol) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) gstore(slot + (wave * 32 + (r & 3) + 8 * (r >> 2) + rbase) * 128 + t * 32 + col, acc[t][r]);
  }
}

// ---- full tiles: 2 x 2 waves of 2 x 2 MFMA tiles ------------------------------------------------------------------------------
// The loop gemm.h runs per k tile (barrier, registers -> LDS, barrier, fetch, 64 MFMAs with ~50 VALU instructions of
// pointer / LDS-address arithmetic) kept the matrix pipe 0.76 busy at three workgroups per CU; timing ablations
// (profiles/r03_wgrad_ablations.md): operands from cache -4 %, no barriers -6 %, both -7 %: not memory, not the
// barriers alone -- on gfx950 the fp32 MFMA shares the SIMD's ALUs with the VALU, so every address instruction is matrix
// time.  This loop has none: global addresses are a scalar base (advanced on the SALU) + a constant per-lane offset, the LDS
// tile keeps a wave's two 32-column sub-tiles 64 floats apart so that one ds_read2st64_b32 with immediate offsets fetches both
// operands of a k step, and the tile is double-buffered: ONE barrier per k tile, the next tile's registers -> LDS copy and the
// fetch after next issue at the head of the MFMA burst.  64 KB of LDS: two workgroups per CU.
// 16-byte load at (uniform base) + (per-lane 32-bit byte offset) inside the branch-free k loops.  The empty asm only keeps the
// zero-extension of the offset from being hoisted out of the loop as a 64-bit register pair (the address would then need a
// 64-bit VALU add per load); it emits no instruction and hides nothing from the wait-count pass.
__device__ __forceinline__ f32x4 steady_load(const char* base, unsigned& off) {
  asm volatile("" : "+v"(off));
  return gload4u((const float*)(base + off));
}
constexpr int WTILE = GBK * GLDR;                // floats per staged operand tile (16 KB), [k][128] with permuted columns
#ifndef OBJ_WG_WAVES
#define OBJ_WG_WAVES 2
#endif
// position of column c of an operand tile inside its k row: 32-column group g = (wave coordinate w, sub-tile i) = (g >> 1, g & 1)
// sits at 32 (2 i + w)
__device__ __forceinline__ int wg_pos(int c) { const int g = c >> 5; return (((g & 1) << 1) + (g >> 1)) * 32 + (c & 31); }

struct WgOperand {             // one operand of a full tile: 128 columns ("rows" of the staged tile) x 32 k per tile
  const char* base;            // uniform: element (k0, row0) of the k tile to fetch next
  long step;                   // uniform: bytes per k tile
  unsigned off[4];             // this thread's four 16-byte pieces (k = tid / 32 + 8 i, columns 4 (tid % 32) .. + 3): bytes from base
  int cols_left;               // columns of the operand from this thread's first one on (<= 0: none)
  bool fast, aligned;          // uniform: all 128 columns exist; 16-byte aligned pieces
  bool over;                   // uniform: fewer than 128 columns, but a 128-column read of any row that has GBK more rows after
                               // it stays inside the matrix (it runs on into the next rows): the branch-free loop may fetch such
                               // rows whole -- the surplus columns only ever meet output rows / columns that are never read back
  f32x4 v[4];
  __device__ __forceinline__ void init(const float* src, long ld, long row0, long rows, long kbeg, int tid) {
    base = (const char*)(src + kbeg * ld + row0);
    step = GBK * ld * 4;
    cols_left = (int)(rows - row0) - 4 * (tid & 31);
#pragma unroll
    for (int i = 0; i < 4; ++i) off[i] = 4u * (unsigned)(((tid >> 5) + 8 * i) * ld + 4 * (tid & 31));
    fast = row0 + GBN <= rows;
    over = !fast && row0 + GBN <= (long)(GBK + 1) * ld;
    aligned = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
  }
  // the contraction range must be zero-filled in BOTH operands (0 x garbage could be NaN): only a ragged last tile needs it
  // Piece i of the k tile that starts krem points before the end of the slice (32-bit counters: the SALU has no 64-bit ordered
  // compare).  The contraction range must be zero-filled in BOTH operands (0 x garbage could be NaN): a ragged last tile only.
  __device__ __forceinline__ void fetch_piece(int i, int krem, int tid) {
    if (fast && krem >= GBK) {                 // uniform
      v[i] = aligned ? gload4((const float*)(base + off[i])) : gload4u((const float*)(base + off[i]));
    } else {
      const bool kin = (tid >> 5) + 8 * i < krem;
      if (kin && cols_left >= 4) {
        v[i] = gload4u((const float*)(base + off[i]));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][j] = (kin && j < cols_left) ? gload((const float*)(base + off[i]) + j) : 0.f;
      }
    }
  }
  __device__ __forceinline__ void advance() { base += step; }       // after the four pieces of a k tile
  __device__ __forceinline__ void fetch(int krem, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fetch_piece(i, krem, tid);
    advance();
  }
  __device__ __forceinline__ void put_piece(int i, float* tile, int tid) const {
    *(f32x4*)&tile[((tid >> 5) + 8 * i) * GLDR + wg_pos(4 * (tid & 31))] = v[i];
  }
  __device__ __forceinline__ void put(float* tile, int tid) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) put_piece(i, tile, tid);
  }
};
#define OBJ_WG_SYNC() __syncthreads()

__device__ __forceinline__ void wgrad_full_piece(const WgProduct& prv, const WgTile& tl, long kbeg, long kend,
                                                 float* slot, float* lds, int tid, long P) {
  // the lists are read with vector loads (the compiler cannot know they are constant): say that they are uniform
  const WgProduct pr{uni(prv.A), uni(prv.lda), uni(prv.B), uni(prv.ldb), prv.C, prv.ldc, uni(prv.rowsum), uni(prv.M), uni(prv.N),
                     uni(prv.segsum), uni(prv.seg_ld)};
  const long nseg16 = (uni(P) + 15) >> 4;
  kbeg = uni(kbeg); kend = uni(kend); slot = uni(slot);
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const long m0 = (long)uni((int)tl.by) * GBM, n0 = (long)uni((int)tl.bx) * GBN;
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float rsum = 0.f;
  const bool want_rowsum = pr.rowsum != nullptr && tl.bx == 0;
  WgOperand opa, opb;
  opa.init(pr.A, pr.lda, m0, pr.M, kbeg, tid);
  opb.init(pr.B, pr.ldb, n0, pr.N, kbeg, tid);
  const int klen = (int)(kend - kbeg);   // a slice is at most 2^31 points long
  opa.fetch(klen, tid);
  opb.fetch(klen, tid);
  opa.put(lds, tid);
  opb.put(lds + WTILE, tid);
  if (klen > GBK) {
    opa.fetch(klen - GBK, tid);
    opb.fetch(klen - GBK, tid);
  }
  const int half = lane >> 5, rl = lane & 31;
  const int aoff = 16 * half * GLDR + wm * 32 + rl, boff = 16 * half * GLDR + wn * 32 + rl;
  const int roff = (tid >> 7) * 16 * GLDR + wg_pos(tid & 127);
  int buf = 0;
  // One k tile.  STEADY: both operands are full-width panels and at least two more full k tiles follow -- the body has no
  // branch, so the compiler's wait counts stay exact (a load waits for its own piece only: vmcnt(6), not vmcnt(0)).
  auto k_tile = [&](auto steady, int krem) __attribute__((always_inline)) {
    constexpr bool STEADY = decltype(steady)::value;
    OBJ_WG_SYNC();                       // this tile is complete in LDS; every wave is done reading the other buffer
    const float* As = lds + buf * 2 * WTILE;
    const float* Bs = As + WTILE;
    float* An = lds + (buf ^ 1) * 2 * WTILE;
    float* Bn = An + WTILE;
    if (want_rowsum) { // thread -> row tid % 128, 16 of the 32 k
      float s16 = 0.f;                   // (rsum keeps its own chain of additions: the bias gradients stay bit-equal)
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) { const float x = As[roff + kk * GLDR]; rsum += x; s16 += x; }
      if (pr.segsum) {                   // uniform
        const long seg = ((kbeg + (klen - krem)) >> 4) + (tid >> 7);
        if (seg < nseg16 && m0 + (tid & 127) < pr.M) gstore(pr.segsum + seg * pr.seg_ld + m0 + (tid & 127), s16);
      }
    }
    // MFMA step (s4, s) of lane half h contracts k = 16 h + 4 s4 + s; the fragments of group s4 + 1 are read before the 16
    // MFMAs of group s4 issue
    float fa[2][2][4], fb[2][2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      fa[0][0][s] = As[aoff + s * GLDR]; fa[0][1][s] = As[aoff + s * GLDR + 64];
      fb[0][0][s] = Bs[boff + s * GLDR]; fb[0][1][s] = Bs[boff + s * GLDR + 64];
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      // a quarter of the staging per MFMA group: piece s4 of the next tile goes registers -> the other LDS buffer, then piece
      // s4 of the tile after it global -> the same registers (one k tile = 64 MFMAs per wave to arrive).  Issued as one burst
      // at the head of the k tile, the workgroups' 32 KiB each queue at the CU's 64 B/clk vector-memory path and the issuing
      // waves stall behind it: +17 % kernel time even with every load hitting in cache (profiles/r03_wgrad_ablations.md)
      if constexpr (STEADY) {
        // Compiler-visible loads (round 5; rounds 3-4 issued them from inline asm into "=v" outputs and waited with a hand-counted
        // s_waitcnt one k tile later -- a copy inserted between issue and wait would have read stale registers).  The compiler's
        // own wait-count pass finds the exact count in this branch-free body: eight loads in flight, in-order return, so the copy
        // of piece s4 waits with vmcnt(6) for the two loads issued for it one k tile ago (checked in the ISA).  steady_load keeps
        // the scalar-base addressing form (global_load_dwordx4 v, v_off, s[base:base+1]): no per-load VALU address arithmetic.
        opa.put_piece(s4, An, tid);
        opb.put_piece(s4, Bn, tid);
        opa.v[s4] = steady_load(opa.base, opa.off[s4]);
        opb.v[s4] = steady_load(opb.base, opb.off[s4]);
      } else if (krem > GBK) {           // uniform
        opa.put_piece(s4, An, tid);
        opb.put_piece(s4, Bn, tid);
        if (krem > 2 * GBK) {
          opa.fetch_piece(s4, krem - 2 * GBK, tid);
          opb.fetch_piece(s4, krem - 2 * GBK, tid);
        }
      }
      if (s4 < 3) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int kk = 4 * (s4 + 1) + s;
          fa[(s4 + 1) & 1][0][s] = As[aoff + kk * GLDR]; fa[(s4 + 1) & 1][1][s] = As[aoff + kk * GLDR + 64];
          fb[(s4 + 1) & 1][0][s] = Bs[boff + kk * GLDR]; fb[(s4 + 1) & 1][1][s] = Bs[boff + kk * GLDR + 64];
        }
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
    if (STEADY || krem > 2 * GBK) { opa.advance(); opb.advance(); }
    buf ^= 1;
  };
  int krem = klen;
  // a panel with fewer than 128 columns (the 104-column object voxel embedding, the 64 rows of the object direction layer) used to
  // take the branchy loop for its whole slice; it now leaves the branch-free loop one k tile earlier instead (WgOperand::over)
  const int steady_min = (opa.fast && opb.fast ? 3 : 4) * GBK;
  if ((opa.fast || opa.over) && (opb.fast || opb.over) && krem >= steady_min) {       // uniform
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the second tile is in registers (and the compiler knows it)
    for (; krem >= steady_min; krem -= GBK) k_tile(std::true_type{}, krem);
  }
  for (; krem > 0; krem -= GBK) k_tile(std::false_type{}, krem);
  __syncthreads();                       // the row-sum exchange below reuses the LDS buffers
  if (want_rowsum) {                     // the two 16-k halves of a row live in threads tid and tid + 128 (fixed order)
    if (tid >= 128) lds[tid - 128] = rsum;
    __syncthreads();
    if (tid < 128) gstore(slot + 128 * 128 + tid, rsum + lds[tid]);
  }
  // D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); the partial tile goes to the unit's slot
  // as a dense 128 x 128 block (rows / columns past the product's extent are never read back)
  const int col = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = t >> 1, j = t & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      gstore(slot + (wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rbase) * 128 + wn * 64 + j * 32 + col, acc[t][r]);
  }
}

// ---- 256 x 256 tiles: one workgroup per CU, 2 x 2 waves of 4 x 4 MFMA tiles ------------------------------------------------
// The full-tile kernel above is held by the part's power management, not by its instruction stream (profiles/r03_train_pmc.md:
// busy x clock did not move when its issue overhead halved): ~3.5 TB/s of operand panels stream in next to the MFMAs.  A
// 256 x 256 output tile moves HALF the bytes per MFMA -- from memory (each panel of a 256 x 256 product is fetched once instead
// of twice), into LDS, and out of it (the wave tile is 128 x 128: 4 + 4 operand registers feed 16 MFMAs, against 2 + 2 for 4) --
// at the price of the inference kernel's structure: 256 accumulator registers per lane, one wave per SIMD, 128 KB of LDS.
// Same staging scheme as wgrad_full_piece: scalar base + constant lane offsets, a [k][256] tile with a wave's four 32-column
// sub-tiles 64 floats apart (ds_read2st64_b32 with immediate offsets), double-buffered LDS with one barrier per k tile, one of
// the sixteen 16-byte pieces per MFMA step (registers -> other buffer, then global -> the same registers, waited for with
// vmcnt(15) one k tile later).  The partial tile goes to the slots of its four 128 x 128 quadrants (each wave owns one), so
// wgrad_fixup_kernel does not know the difference.  Measured: 0.050 ms per launch and 128 x 128 tile against 0.065 for the
// kernel above, 0.90 of the MFMA-only ceiling at the clock the part holds (profiles/r03_wgrad_ablations.md, section 4).
constexpr int BTILE = GBK * 256;                 // floats per staged operand tile (32 KB)
__device__ __forceinline__ int big_pos(int c) { const int g = c >> 5; return (((g & 3) << 1) + (g >> 2)) * 32 + (c & 31); }

struct BigOperand {            // one operand of a 256 x 256 tile: 256 columns x 32 k per tile, all columns exist
  const char* base;            // uniform: element (k0, row0) of the k tile to fetch next
  long step;                   // uniform: bytes per k tile
  unsigned off[8];             // piece i: k = tid / 64 + 4 i, columns 4 (tid % 64) .. + 3: bytes from base
  f32x4 v[8];
  __device__ __forceinline__ void init(const float* src, long ld, long row0, long kbeg, int tid) {
    base = (const char*)(src + kbeg * ld + row0);
    step = GBK * ld * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) off[i] = 4u * (unsigned)(((tid >> 6) + 4 * i) * ld + 4 * (tid & 63));
  }
  // compiler-tracked form (prologue and the last tiles of a slice); rows of the k tile past the slice end are zero-filled
  __device__ __forceinline__ void fetch_piece(int i, int krem, int tid) {
    const bool kin = (tid >> 6) + 4 * i < krem;            // wave-uniform
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    v[i] = kin ? gload4u((const float*)(base + off[i])) : z;
  }
  __device__ __forceinline__ void fetch(int krem, int tid) {
#pragma unroll
    for (int i = 0; i < 8; ++i) fetch_piece(i, krem, tid);
    base += step;
  }
  __device__ __forceinline__ void put_piece(int i, float* tile, int tid) const {
    *(f32x4*)&tile[((tid >> 6) + 4 * i) * 256 + big_pos(4 * (tid & 63))] = v[i];
  }
  __device__ __forceinline__ void put(float* tile, int tid) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) put_piece(i, tile, tid);
  }
};

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) wgrad_big_kernel(const WgradArgs* __restrict__ ap) {
  __shared__ __attribute__((aligned(16))) float lds[4 * BTILE];
  const WgradArgs& a = *ap;
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const long P = uni(a.P);
  const long KT = (P + GBK - 1) / GBK;
  const int nz = uni(a.nz), nzb = uni(a.nzb);              // (nz: the slot stride; nzb: this kernel's slices per tile)
  const long L = (KT + nzb - 1) / nzb;                     // k iterations per slice
  const int b = uni((int)(blockIdx.x / (unsigned)nzb)), z = uni((int)(blockIdx.x % (unsigned)nzb));   // product by product, slice by slice
  const int t0 = 4 * b;                                    // first quadrant's tile
  const WgProduct& prv = a.prod[a.tile[t0].prod];
  const float* A = uni(prv.A); const float* B = uni(prv.B);
  const long lda = uni(prv.lda), ldb = uni(prv.ldb);
  const bool want_rowsum = uni(prv.rowsum) != nullptr;
  const long kbeg = (long)z * L * GBK;
  long kend = kbeg + L * GBK;
  if (kend > P) kend = P;
  const int klen = (int)(kend - kbeg);

  f32x16 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float rsum = 0.f;
  BigOperand opa, opb;
  opa.init(A, lda, 0, kbeg, tid);
  opb.init(B, ldb, 0, kbeg, tid);
  opa.fetch(klen, tid);
  opb.fetch(klen, tid);
  opa.put(lds, tid);
  opb.put(lds + BTILE, tid);
  if (klen > GBK) {
    opa.fetch(klen - GBK, tid);
    opb.fetch(klen - GBK, tid);
  }
  const int half = lane >> 5, rl = lane & 31;
  const int aoff = 16 * half * 256 + wm * 32 + rl, boff = 16 * half * 256 + wn * 32 + rl;
  const int roff = big_pos(tid);                           // row sums: thread = row, all 32 k of a tile in ascending order
  int buf = 0;
  auto k_tile = [&](auto steady, int krem) __attribute__((always_inline)) {
    constexpr bool STEADY = decltype(steady)::value;
    __syncthreads();                     // this tile is complete in LDS; every wave is done reading the other buffer
    const float* As = lds + buf * 2 * BTILE;
    const float* Bs = As + BTILE;
    float* An = lds + (buf ^ 1) * 2 * BTILE;
    float* Bn = An + BTILE;
    if (want_rowsum) {
#pragma unroll
      for (int kk = 0; kk < GBK; ++kk) rsum += As[roff + kk * 256];
    }
    float fa[2][4], fb[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { fa[0][i] = As[aoff + i * 64]; fb[0][i] = Bs[boff + i * 64]; }
#pragma unroll
    for (int q = 0; q < 16; ++q) {       // MFMA step q = 4 s4 + s of lane half h contracts k = 16 h + q
      // staging piece q (A pieces 0..7 on the even steps, B pieces on the odd ones)
      if constexpr (STEADY) {
        // sixteen loads in flight, in-order return: the compiler waits with vmcnt(15) for this piece's (see wgrad_full_piece)
        if ((q & 1) == 0) {
          opa.put_piece(q >> 1, An, tid);
          opa.v[q >> 1] = steady_load(opa.base, opa.off[q >> 1]);
        } else {
          opb.put_piece(q >> 1, Bn, tid);
          opb.v[q >> 1] = steady_load(opb.base, opb.off[q >> 1]);
        }
      } else if (krem > GBK) {           // uniform
        if ((q & 1) == 0) {
          opa.put_piece(q >> 1, An, tid);
          if (krem > 2 * GBK) opa.fetch_piece(q >> 1, krem - 2 * GBK, tid);
        } else {
          opb.put_piece(q >> 1, Bn, tid);
          if (krem > 2 * GBK) opb.fetch_piece(q >> 1, krem - 2 * GBK, tid);
        }
      }
      if (q < 15) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fa[(q + 1) & 1][i] = As[aoff + (q + 1) * 256 + i * 64];
          fb[(q + 1) & 1][i] = Bs[boff + (q + 1) * 256 + i * 64];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[4 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1][i], fb[q & 1][j], acc[4 * i + j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (STEADY || krem > 2 * GBK) { opa.base += opa.step; opb.base += opb.step; }
    buf ^= 1;
  };
  int krem = klen;
  if (krem >= 3 * GBK) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the second tile is in registers (and the compiler knows it)
    for (; krem >= 3 * GBK; krem -= GBK) k_tile(std::true_type{}, krem);
  }
  for (; krem > 0; krem -= GBK) k_tile(std::false_type{}, krem);

  // every wave owns one quadrant = one 128 x 128 tile of the list; its partial tile goes to that tile's slot of this slice
  float* slot = uni(a.partials) + ((long)(t0 + wm * 2 + wn) * nz + z) * kWgradSlotFloats;
  if (want_rowsum)                       // thread = row: rows 0..127 belong to quadrant (0,0)'s tile, rows 128..255 to (1,0)'s
    gstore(uni(a.partials) + ((long)(t0 + 2 * (tid >> 7)) * nz + z) * kWgradSlotFloats + 128 * 128 + (tid & 127), rsum);
  const int col = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int i = t >> 2, j = t & 3;
#pragma unroll
    for (int r = 0; r < 16; ++r) gstore(slot + (i * 32 + (r & 3) + 8 * (r >> 2) + rbase) * 128 + j * 32 + col, acc[t][r]);
  }
}

// The lists travel as kernel arguments (no device copy to enqueue, no allocation) and are parked in device memory by this
// one-workgroup kernel: indexing a by-value argument struct with a run-time index would make every kernel that does it
// keep a private copy of the 2.5 KB struct in scratch memory.
__global__ void __launch_bounds__(256) wgrad_list_kernel(const WgradArgs a, WgradArgs* __restrict__ out) {
  const unsigned* src = (const unsigned*)&a;
  unsigned* dst = (unsigned*)out;
  for (unsigned i = threadIdx.x; i < sizeof(WgradArgs) / 4; i += 256) dst[i] = src[i];
}

// One workgroup per UNIT = (tile, k slice).  TAIL = false: the full tiles of the list ([0, nfull)), TAIL = true: the
// ragged ones ([nfull, ntile)), one launch each (one code path per kernel keeps it at 3 workgroups per CU).  Units are
// numbered product by product, inside a product slice by slice, inside a slice tile by tile: workgroups that are resident
// together contract the SAME points for the neighbouring tiles of one product, so each 32-point panel of dY / X is
// fetched from HBM once and re-read from L2 (a tile-major order -- every workgroup a different k range of one tile --
// streamed the operands 4-6 times: measured HBM-bound).  Every unit leaves its partial tile in its own slot.
template <bool TAIL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TAIL ? 3 : OBJ_WG_WAVES, TAIL ? 3 : OBJ_WG_WAVES))) wgrad_units_kernel(const WgradArgs* __restrict__ ap) {
  __shared__ __attribute__((aligned(16))) float lds[TAIL ? 2 * GTILE : 4 * WTILE];
  const WgradArgs& a = *ap;
  const int tid = threadIdx.x;
  const long P = uni(a.P);
  const long KT = (P + GBK - 1) / GBK;
  const int nz = uni(a.nz), nzk = uni(TAIL ? a.nzr : a.nzu);      // (nz: the slot stride; nzk: this kernel's slices per tile)
  const long L = (KT + nzk - 1) / nzk;                     // k iterations per slice
  const int t0 = TAIL ? a.nfull : a.nbig, t1 = TAIL ? a.ntile : a.nfull;
  // unit -> (tile, slice): walk the runs of tiles that belong to one product.  Workgroups are dealt round-robin to the 8
  // XCDs; with a.xcd every XCD takes one contiguous eighth of the units, so the tiles of one slice (which read the same
  // operand panels) run on ONE XCD and share its L2
  long u = blockIdx.x;
  // Issue priority (OBJNERF_WGRAD_PRIO, a.xcd >> 1): the three workgroups of a CU run the same loop with the same period; with
  // equal priority they share the matrix pipe round-robin, reach their barriers together and leave the pipe idle through the
  // staging phase (a convoy).  Distinct priorities order them: the first never waits for the pipe, the others fill its gaps.
  // 1: priority = hardware wave slot (HW_ID.wave_id: the co-resident waves of a SIMD have distinct slots), 2: a hash of the unit
  const int prio_mode = (a.xcd >> 1) & 3;
  if (prio_mode) {
    const unsigned slot = (prio_mode == 1 ? __builtin_amdgcn_s_getreg(6148) : ((blockIdx.x * 2654435761u) >> 16)) % 3u;
    if (slot == 0) __builtin_amdgcn_s_setprio(2);
    else if (slot == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
  }
  if (a.xcd & 1) {
    const long per = (gridDim.x + 7) / 8;
    u = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (u >= gridDim.x) return;
  }
  int t = t0, z = 0;
  while (t < t1) {
    int run = 1;
    while (t + run < t1 && a.tile[t + run].prod == a.tile[t].prod) ++run;
    if (u < (long)run * nzk) { z = (int)(u / run); t += (int)(u % run); break; }
    u -= (long)run * nzk;
    t += run;
  }
  if (t >= t1) return;
  t = uni(t); z = uni(z);
  const WgTile tl = a.tile[t];
  const long kbeg = (long)z * L * GBK;
  long kend = kbeg + L * GBK;
  if (kend > P) kend = P;
  float* slot = a.partials + ((long)t * nz + z) * kWgradSlotFloats;
  if constexpr (TAIL) wgrad_tail_piece(a.prod[tl.prod], tl, kbeg, kend, slot, lds, tid, P);      // an empty slice writes zeros
  else wgrad_full_piece(a.prod[tl.prod], tl, kbeg, kend, slot, lds, tid, P);
}

// Adds a tile's slices to dW in ascending slice order (= ascending points): every bit of the result is reproducible.
// grid (tile, eighth of the tile's 16,384 elements)
__global__ void __launch_bounds__(256) wgrad_fixup_kernel(const WgradArgs* __restrict__ ap) {
  const WgradArgs& a = *ap;
  const int t = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
  const int nz = a.nz, nzt = t < a.nbig ? a.nzb : (t < a.nfull ? a.nzu : a.nzr);          // slot stride; slices this tile was cut into
  const WgTile tl = a.tile[t];
  const WgProduct pr = a.prod[tl.prod];
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  float rs = 0.f;
  const bool want_rs = part == 0 && pr.rowsum && tl.bx == 0 && tid < 128;
  // eight slices' values are fetched before they are added, in the same ascending order (the same bits): one slice at a time the
  // loop was a chain of up to 153 dependent round trips per thread (round 6: step -0.19 ms, profiles/r06_train_ab_fixup.txt)
  int z = 0;
  for (; z + 8 <= nzt; z += 8) {
    float v[8][8], r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* slot = a.partials + ((long)t * nz + z + u) * kWgradSlotFloats;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[u][i] = gload(slot + (part * 8 + i) * 256 + tid);
      r[u] = want_rs ? gload(slot + 128 * 128 + tid) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += v[u][i];
      if (want_rs) rs += r[u];
    }
  }
  for (; z < nzt; ++z) {
    const float* slot = a.partials + ((long)t * nz + z) * kWgradSlotFloats;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += gload(slot + (part * 8 + i) * 256 + tid);
    if (want_rs) rs += gload(slot + 128 * 128 + tid);
  }
  const long m0 = (long)tl.by * GBM, n0 = (long)tl.bx * GBN;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = (part * 8 + i) * 256 + tid, ml = e >> 7, nl = e & 127;
    if (m0 + ml < pr.M && n0 + nl < pr.N && nl < 32 * (int)tl.ncol) {
      float* c = pr.C + (m0 + ml) * pr.ldc + n0 + nl;
      gstore(c, gload(c) + acc[i]);
    }
  }
  if (want_rs && m0 + tid < pr.M) gstore(pr.rowsum + m0 + tid, gload(pr.rowsum + m0 + tid) + rs);
}

// ---- the 1- and 3-row heads on the VALU ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) heads_wgrad_kernel(const HeadArgs a) {
  // the chunk's dY rows (<= 12 KB) are staged in LDS once: every thread needs every one of them (uniform reads = LDS
  // broadcasts), and the bias sums walk them without a global-memory round trip per point (1024 dependent-latency loads by
  // three threads were most of this kernel's 0.25 ms)
  __shared__ float dys[kHeadChunk * 3];
  const HeadItem h = a.h[blockIdx.y];
  const int tid = threadIdx.x;
  const long p0 = (long)blockIdx.x * kHeadChunk, p1 = p0 + kHeadChunk < a.P ? p0 + kHeadChunk : a.P;
  const int np = (int)(p1 - p0), no = h.no;
  for (int i = tid; i < np * no; i += 256) dys[i] = gload(h.dY + p0 * no + i);
  __syncthreads();
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, b = 0.f;
  // thread = input column; 16 points per trip so that 16 row loads are in flight (the loop is latency-bound otherwise);
  // the sums stay sequential in p: fixed order
  if (tid < h.ni) {
    const float* x = h.X + p0 * h.ldx + tid;
    int p = 0;
    for (; p + 16 <= np; p += 16) {
      float xv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) xv[u] = gload(x + (long)(p + u) * h.ldx);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float* dy = dys + (p + u) * no;
        s0 += dy[0] * xv[u];
        if (no > 1) { s1 += dy[1] * xv[u]; s2 += dy[2] * xv[u]; }
      }
    }
    for (; p < np; ++p) {
      const float xv = gload(x + (long)p * h.ldx);
      s0 += dys[p * no] * xv;
      if (no > 1) { s1 += dys[p * no + 1] * xv; s2 += dys[p * no + 2] * xv; }
    }
  }
  if (tid < no) {                        // (sixteen LDS reads in flight per trip, added in ascending order: the same bits)
    int p = 0;
    for (; p + 16 <= np; p += 16) {
      float t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = dys[(p + u) * no + tid];
#pragma unroll
      for (int u = 0; u < 16; ++u) b += t[u];
    }
    for (; p < np; ++p) b += dys[p * no + tid];
  }
  float* slot = a.partials + ((long)blockIdx.y * a.nchunks + blockIdx.x) * kHeadSlotFloats;
  gstore(slot + tid, s0); gstore(slot + 256 + tid, s1); gstore(slot + 512 + tid, s2);
  if (tid < 3) gstore(slot + 768 + tid, b);
}
__global__ void __launch_bounds__(256) heads_fixup_kernel(const HeadArgs a) {
  const HeadItem h = a.h[blockIdx.x];
  const int tid = threadIdx.x;
  float s[3] = {0.f, 0.f, 0.f}, b = 0.f;
  // ascending chunks = ascending points; sixteen chunks' partial sums are fetched before they are added (same order of additions)
  int c = 0;
  for (; c + 16 <= a.nchunks; c += 16) {
    float v[16][3], bv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float* slot = a.partials + ((long)blockIdx.x * a.nchunks + c + u) * kHeadSlotFloats;
      v[u][0] = gload(slot + tid); v[u][1] = gload(slot + 256 + tid); v[u][2] = gload(slot + 512 + tid);
      bv[u] = tid < 3 ? gload(slot + 768 + tid) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) { s[0] += v[u][0]; s[1] += v[u][1]; s[2] += v[u][2]; b += bv[u]; }
  }
  for (; c < a.nchunks; ++c) {
    const float* slot = a.partials + ((long)blockIdx.x * a.nchunks + c) * kHeadSlotFloats;
    s[0] += gload(slot + tid); s[1] += gload(slot + 256 + tid); s[2] += gload(slot + 512 + tid);
    if (tid < 3) b += gload(slot + 768 + tid);
  }
  if (tid < h.ni)
    for (int o = 0; o < h.no; ++o) h.dW[o * h.ldw + tid] += s[o];
  if (h.db && tid < h.no) h.db[tid] += b;
}

// ---- host ---------------------------------------------------------------------------------------------------------
static long wgrad_base_slices(long P) {
  static const long iters = [] {
    const char* e = getenv("OBJNERF_WGRAD_KITERS");
    const long v = e ? atol(e) : 0;
    return v > 0 ? v : (long)kWgradSliceIters;
  }();
  const long KT = (P + GBK - 1) / GBK;
  const long nz = (KT + iters - 1) / iters;
  return nz < 1 ? 1 : nz;
}
// upper bound (scratch sizing): the launch may cut up to a quarter more slices than the base count, see wgrad_pick_slices
int wgrad_slices(long P) {
  const long nz = wgrad_base_slices(P) * 5 / 4 + 1;
  return (int)(nz > kWgradMaxSlices ? kWgradMaxSlices : nz);
}
// Slices per tile for a pass with `nbig` 256 x 256 tiles and `nfull` ordinary full tiles: the big-tile workgroups run one per CU
// (a unit = four tiles' work on a whole CU), the ordinary ones two per CU (a unit = one tile on half a CU: half the time), both
// with every resident slot busy until their last round -- the base count (64 k tiles per slice) leaves e.g. 53 x 128 units =
// 13.25 rounds of 512 slots, the last round three quarters empty.  Take the count in [base, 1.25 base] that costs the fewest
// rounds per unit of work (a function of P, the model's tile list and the device only: the sums stay reproducible run to run).
static int wgrad_pick_slices(long P, int nbig, int nfull) {
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  static const bool fixed = getenv("OBJNERF_WGRAD_KITERS") != nullptr;       // tuning runs keep the count they ask for
  const long base = wgrad_base_slices(P), cap = wgrad_slices(P);
  if (fixed || nbig + nfull <= 0) return (int)(base > cap ? cap : base);
  long best = base > cap ? cap : base;
  double best_cost = 1e300;
  for (long nz = best; nz <= cap; ++nz) {
    const long rb = (nz * nbig + cus - 1) / cus, rf = (nz * nfull + 2 * cus - 1) / (2 * cus);
    const double cost = (2.0 * (double)rb + (double)rf) / (double)nz;
    if (cost < best_cost - 1e-12) { best_cost = cost; best = nz; }
  }
  return (int)best;
}

void WgradBatch::add(const float* dY, long lddy, const float* X, long ldx, int out, int in, float* dW, long ldw, float* db,
                     float* segsum, long seg_ld) {
  if (a.nprod >= kWgradMaxProducts) { overflow = true; return; }
  // 256 x 256 tiles (wgrad_big_kernel) for the products with exactly two row tiles and at least two full column tiles:
  // OBJNERF_WGRAD_BIG=0 keeps every full tile in the 128 x 128 kernel (developer A/B switch)
  static const bool big_on = [] { const char* e = getenv("OBJNERF_WGRAD_BIG"); return !e || atoi(e) != 0; }();
  const bool big = big_on && out == 2 * GBM && in >= 2 * GBN;
  if (segsum && (big || !db)) { overflow = true; return; }     // segment sums ride on the rowsum tile of the 128 x 128 kernels
  const int pi = a.nprod++;
  a.prod[pi] = WgProduct{dY, lddy, X, ldx, dW, ldw, db, out, in, segsum, seg_ld};
  const int ny = (out + GBM - 1) / GBM, nx = (in + GBN - 1) / GBN;
  for (int by = 0; by < ny; ++by)
    for (int bx = 0; bx < nx; ++bx) {
      if (a.ntile >= kWgradMaxTiles) { overflow = true; return; }
      const int cols = in - bx * GBN < GBN ? in - bx * GBN : GBN;
      const int sub = (cols + 31) / 32;                          // live 32-column sub-tiles
      const WgTile tl{(unsigned char)pi, (unsigned char)by, (unsigned char)bx, (unsigned char)(sub >= 4 ? 4 : sub)};
      // list order: quadrants of the 256 x 256 tiles | other full tiles | ragged tiles; insertion keeps each part in call order,
      // so the quadrants of one product sit together as (0,0) (0,1) (1,0) (1,1)
      int at = a.ntile;
      if (tl.ncol == 4) at = (big && bx < 2) ? a.nbig : a.nfull;
      for (int i = a.ntile; i > at; --i) a.tile[i] = a.tile[i - 1];
      a.tile[at] = tl;
      ++a.ntile;
      if (tl.ncol == 4) { ++a.nfull; if (big && bx < 2) ++a.nbig; }
    }
}
void WgradBatch::add_head(const float* dY, int no, const float* X, long ldx, int ni, float* dW, long ldw, float* db) {
  if (h.nheads >= kMaxHeads || no > 3 || ni > 256) { overflow = true; return; }
  h.h[h.nheads++] = HeadItem{dY, X, dW, db, ldx, ldw, no, ni};
}
// slices: 0 = the pass's own choice (wgrad_pick_slices); > 0: that many k slices per tile (the caller has sized the slot area for
// ntile * slices slots) -- the small per-ray pass of train.hip, whose few tiles need many short slices to fill the device
int WgradBatch::launch(long P, float* scratch, hipStream_t s, int slices) {
  if (overflow) return set_error(-3, "wgrad: work list overflow");
  if (a.ntile > kWgradSlotTiles) return set_error(-3, "wgrad: more output tiles than the scratch is sized for (kWgradSlotTiles)");
  if (P <= 0) return 0;
  a.P = P;
  a.partials = scratch;
  if (a.ntile > 0) {
    static_assert(sizeof(WgradArgs) % 4 == 0 && sizeof(WgradArgs) <= 4096, "the lists travel as kernel arguments");
    WgradArgs* dev = (WgradArgs*)(scratch + wgrad_scratch_floats(P) - kWgradListFloats);
    const long KTl = (P + GBK - 1) / GBK;
    // ONE round of workgroups per kernel (round 6) instead of the ~4-6 rounds the 64-k-tile slices make at the reference batch:
    // cus / tiles slices per 256 x 256 tile (one workgroup per CU), 2 cus / tiles per 128 x 128 tile, 3 cus / tiles per ragged tile
    // -- a quarter of the partial tiles to write and to add up again, one prologue / epilogue per resident workgroup, no
    // part-empty last round.  Workgroups that run together still contract the same points for the neighbouring tiles of a
    // product (the launch order is unchanged).  OBJNERF_WGRAD_ROUNDS=many: every tile cut into 64-k-tile slices as in rounds 3-5;
    // =big: only the 256 x 256 tiles in one round (this round's first step).
    static const int rounds_mode = [] {
      const char* e = getenv("OBJNERF_WGRAD_ROUNDS");
      const char* old = getenv("OBJNERF_WGRAD_BIG_ROUNDS");
      if ((e && !strcmp(e, "many")) || (old && !strcmp(old, "many"))) return 0;
      return (e && !strcmp(e, "big")) ? 1 : 2;
    }();
    static const int cus_n = [] {
      int dev = 0, n = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
      return n > 0 ? n : 256;
    }();
    const int nbt = a.nbig / 4, nut = a.nfull - a.nbig, nrt = a.ntile - a.nfull;
    auto one_round = [&](int slots, int tiles) {          // slices per tile that fill `slots` resident workgroups once
      if (tiles <= 0) return 1;
      long v = slots / tiles;
      const long cap = wgrad_slices(P) < KTl ? wgrad_slices(P) : KTl;      // (the slot area is sized for wgrad_slices(P) per tile)
      if (v > cap) v = cap;
      return (int)(v < 1 ? 1 : v);
    };
    int nzb, nzu, nzr;
    if (slices > 0) nzb = nzu = nzr = (int)(slices < KTl ? slices : KTl);
    else if (rounds_mode == 0) nzb = nzu = nzr = wgrad_pick_slices(P, nbt, nut);
    else if (rounds_mode == 1) { nzu = nzr = wgrad_pick_slices(P, 0, nut); nzb = one_round(cus_n, nbt); if (nzb > nzu) nzb = nzu; }
    else { nzb = one_round(cus_n, nbt); nzu = one_round(2 * cus_n, nut); nzr = one_round(3 * cus_n, nrt); }
    a.nz = nzb > nzu ? (nzb > nzr ? nzb : nzr) : (nzu > nzr ? nzu : nzr);       // the slot stride
    a.nzb = nzb; a.nzu = nzu; a.nzr = nzr;
    static const int xcd = [] { const char* e = getenv("OBJNERF_WGRAD_XCD"); return e ? atoi(e) : 0; }();
    static const int prio = [] { const char* e = getenv("OBJNERF_WGRAD_PRIO"); return e ? atoi(e) : 0; }();
    a.xcd = (xcd & 1) | ((prio & 3) << 1);
    hipLaunchKernelGGL(wgrad_list_kernel, dim3(1), dim3(256), 0, s, a, dev);
    if (a.nbig > 0) hipLaunchKernelGGL(wgrad_big_kernel, dim3((a.nbig / 4) * nzb), dim3(256), 0, s, (const WgradArgs*)dev);
    if (a.nfull > a.nbig) hipLaunchKernelGGL(wgrad_units_kernel<false>, dim3((a.nfull - a.nbig) * nzu), dim3(256), 0, s, (const WgradArgs*)dev);
    if (a.ntile > a.nfull) hipLaunchKernelGGL(wgrad_units_kernel<true>, dim3((a.ntile - a.nfull) * nzr), dim3(256), 0, s, (const WgradArgs*)dev);
    hipLaunchKernelGGL(wgrad_fixup_kernel, dim3(a.ntile, 8), dim3(256), 0, s, (const WgradArgs*)dev);
  }
  if (h.nheads > 0) {
    h.P = P;
    h.nchunks = (int)((P + kHeadChunk - 1) / kHeadChunk);
    h.partials = scratch + wgrad_slot_floats(P);
    hipLaunchKernelGGL(heads_wgrad_kernel, dim3(h.nchunks, h.nheads), dim3(256), 0, s, h);
    hipLaunchKernelGGL(heads_fixup_kernel, dim3(h.nheads), dim3(256), 0, s, h);
  }
  return check_launch("wgrad");
}

}  // namespace objnerf
