// wgrad.hip -- grouped, stream-K, deterministic weight gradients of one backward pass (wgrad.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "wgrad.h"
#include "host_api.h"

namespace objnerf {

template <bool TAIL>
__device__ __forceinline__ void wgrad_piece(const WgProduct& pr, const WgTile& tl, long kbeg, long kend,
                                            float* slot, float* lds, int tid) {
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const long m0 = (long)tl.by * GBM, n0 = (long)tl.bx * GBN;
  const int ncol = TAIL ? tl.ncol : 4;
  f32x16 acc[TAIL ? 3 : 4];
#pragma unroll
  for (int i = 0; i < (TAIL ? 3 : 4); ++i)
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
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) rsum += As[((tid >> 7) * 16 + kk) * GLDR + (tid & 127)];
    }
    if constexpr (TAIL) {
      if (m0 + wave * 32 < pr.M) {       // wave-uniform: this wave's row sub-tile exists
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
    } else {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {   // 4 MFMA steps per fragment read
        f32x4 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = GemmOperand<false>::frag(As, wm * 64 + i * 32 + rl, half, s4);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = GemmOperand<false>::frag(Bs, wn * 64 + j * 32 + rl, half, s4);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[2 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[2 * i + j], 0, 0, 0);
      }
    }
  }
  __syncthreads();                       // the next piece's first tile overwrites the LDS buffers
  // row sums: the two 16-k halves of a row live in threads tid and tid + 128: combine through LDS (fixed order)
  if (want_rowsum) {
    if (tid >= 128) lds[tid - 128] = rsum;
    __syncthreads();
    if (tid < 128) slot[128 * 128 + tid] = rsum + lds[tid];
    __syncthreads();
  }
  // D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); the partial tile goes to the unit's slot
  // as a dense 128 x 128 block (rows / columns past the product's extent are never read back)
  const int col = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int t = 0; t < (TAIL ? 3 : 4); ++t) {
    const int i = t >> 1, j = t & 1;
    if (TAIL && t >= ncol) continue;
    const int nl = TAIL ? t * 32 + col : wn * 64 + j * 32 + col;           // column inside the tile
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ml = (TAIL ? wave * 32 : wm * 64 + i * 32) + (r & 3) + 8 * (r >> 2) + rbase;
      slot[ml * 128 + nl] = acc[t][r];
    }
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
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) wgrad_units_kernel(const WgradArgs* __restrict__ ap) {
  __shared__ __attribute__((aligned(16))) float lds[2 * GTILE];
  const WgradArgs& a = *ap;
  const int tid = threadIdx.x;
  const long KT = (a.P + GBK - 1) / GBK;
  const int nz = a.nz;
  const long L = (KT + nz - 1) / nz;                       // k iterations per slice
  const int t0 = TAIL ? a.nfull : 0, t1 = TAIL ? a.ntile : a.nfull;
  // unit -> (tile, slice): walk the runs of tiles that belong to one product.  Workgroups are dealt round-robin to the 8
  // XCDs; with a.xcd every XCD takes one contiguous eighth of the units, so the tiles of one slice (which read the same
  // operand panels) run on ONE XCD and share its L2
  long u = blockIdx.x;
  if (a.xcd) {
    const long per = (gridDim.x + 7) / 8;
    u = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (u >= gridDim.x) return;
  }
  int t = t0, z = 0;
  while (t < t1) {
    int run = 1;
    while (t + run < t1 && a.tile[t + run].prod == a.tile[t].prod) ++run;
    if (u < (long)run * nz) { z = (int)(u / run); t += (int)(u % run); break; }
    u -= (long)run * nz;
    t += run;
  }
  if (t >= t1) return;
  const WgTile tl = a.tile[t];
  const long kbeg = (long)z * L * GBK;
  long kend = kbeg + L * GBK;
  if (kend > a.P) kend = a.P;
  float* slot = a.partials + ((long)t * nz + z) * kWgradSlotFloats;
  wgrad_piece<TAIL>(a.prod[tl.prod], tl, kbeg, kend, slot, lds, tid);      // an empty slice writes zeros
}

// Adds a tile's slices to dW in ascending slice order (= ascending points): every bit of the result is reproducible.
// grid (tile, eighth of the tile's 16,384 elements)
__global__ void __launch_bounds__(256) wgrad_fixup_kernel(const WgradArgs* __restrict__ ap) {
  const WgradArgs& a = *ap;
  const int t = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
  const int nz = a.nz;
  const WgTile tl = a.tile[t];
  const WgProduct pr = a.prod[tl.prod];
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  float rs = 0.f;
  const bool want_rs = part == 0 && pr.rowsum && tl.bx == 0 && tid < 128;
  for (int z = 0; z < nz; ++z) {
    const float* slot = a.partials + ((long)t * nz + z) * kWgradSlotFloats;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += slot[(part * 8 + i) * 256 + tid];
    if (want_rs) rs += slot[128 * 128 + tid];
  }
  const long m0 = (long)tl.by * GBM, n0 = (long)tl.bx * GBN;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = (part * 8 + i) * 256 + tid, ml = e >> 7, nl = e & 127;
    if (m0 + ml < pr.M && n0 + nl < pr.N && nl < 32 * (int)tl.ncol) pr.C[(m0 + ml) * pr.ldc + n0 + nl] += acc[i];
  }
  if (want_rs && m0 + tid < pr.M) pr.rowsum[m0 + tid] += rs;
}

// ---- the 1- and 3-row heads on the VALU ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) heads_wgrad_kernel(const HeadArgs a) {
  const HeadItem h = a.h[blockIdx.y];
  const int tid = threadIdx.x;
  const long p0 = (long)blockIdx.x * kHeadChunk, p1 = p0 + kHeadChunk < a.P ? p0 + kHeadChunk : a.P;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, b = 0.f;
  // thread = input column; 8 points per trip so that 8 row loads are in flight (the loop is latency-bound otherwise);
  // the sums stay sequential in p: fixed order
  if (tid < h.ni) {
    const float* x = h.X + tid;
    long p = p0;
    for (; p + 8 <= p1; p += 8) {
      float xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = x[(p + u) * h.ldx];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float* dy = h.dY + (p + u) * h.no;
        s0 += dy[0] * xv[u];
        if (h.no > 1) { s1 += dy[1] * xv[u]; s2 += dy[2] * xv[u]; }
      }
    }
    for (; p < p1; ++p) {
      const float xv = x[p * h.ldx];
      s0 += h.dY[p * h.no] * xv;
      if (h.no > 1) { s1 += h.dY[p * h.no + 1] * xv; s2 += h.dY[p * h.no + 2] * xv; }
    }
  }
  if (tid < h.no) for (long p = p0; p < p1; ++p) b += h.dY[p * h.no + tid];
  float* slot = a.partials + ((long)blockIdx.y * a.nchunks + blockIdx.x) * kHeadSlotFloats;
  slot[tid] = s0; slot[256 + tid] = s1; slot[512 + tid] = s2;
  if (tid < 3) slot[768 + tid] = b;
}
__global__ void __launch_bounds__(256) heads_fixup_kernel(const HeadArgs a) {
  const HeadItem h = a.h[blockIdx.x];
  const int tid = threadIdx.x;
  float s[3] = {0.f, 0.f, 0.f}, b = 0.f;
  for (int c = 0; c < a.nchunks; ++c) {
    const float* slot = a.partials + ((long)blockIdx.x * a.nchunks + c) * kHeadSlotFloats;
    s[0] += slot[tid]; s[1] += slot[256 + tid]; s[2] += slot[512 + tid];
    if (tid < 3) b += slot[768 + tid];
  }
  if (tid < h.ni)
    for (int o = 0; o < h.no; ++o) h.dW[o * h.ldw + tid] += s[o];
  if (h.db && tid < h.no) h.db[tid] += b;
}

// ---- host ---------------------------------------------------------------------------------------------------------
int wgrad_slices(long P) {
  static const long iters = [] {
    const char* e = getenv("OBJNERF_WGRAD_KITERS");
    const long v = e ? atol(e) : 0;
    return v > 0 ? v : (long)kWgradSliceIters;
  }();
  const long KT = (P + GBK - 1) / GBK;
  const long nz = (KT + iters - 1) / iters;
  return (int)(nz < 1 ? 1 : (nz > kWgradMaxSlices ? kWgradMaxSlices : nz));
}

void WgradBatch::add(const float* dY, long lddy, const float* X, long ldx, int out, int in, float* dW, long ldw, float* db) {
  if (a.nprod >= kWgradMaxProducts) { overflow = true; return; }
  const int pi = a.nprod++;
  a.prod[pi] = WgProduct{dY, lddy, X, ldx, dW, ldw, db, out, in};
  const int ny = (out + GBM - 1) / GBM, nx = (in + GBN - 1) / GBN;
  for (int by = 0; by < ny; ++by)
    for (int bx = 0; bx < nx; ++bx) {
      if (a.ntile >= kWgradMaxTiles) { overflow = true; return; }
      const int cols = in - bx * GBN < GBN ? in - bx * GBN : GBN;
      const int sub = (cols + 31) / 32;                          // live 32-column sub-tiles
      const WgTile tl{(unsigned char)pi, (unsigned char)by, (unsigned char)bx, (unsigned char)(sub >= 4 ? 4 : sub)};
      if (tl.ncol == 4) {                                        // full tiles first, ragged ones behind them
        for (int i = a.ntile; i > a.nfull; --i) a.tile[i] = a.tile[i - 1];
        a.tile[a.nfull++] = tl;
        ++a.ntile;
      } else {
        a.tile[a.ntile++] = tl;
      }
    }
}
void WgradBatch::add_head(const float* dY, int no, const float* X, long ldx, int ni, float* dW, long ldw, float* db) {
  if (h.nheads >= kMaxHeads || no > 3 || ni > 256) { overflow = true; return; }
  h.h[h.nheads++] = HeadItem{dY, X, dW, db, ldx, ldw, no, ni};
}
int WgradBatch::launch(long P, float* scratch, hipStream_t s) {
  if (overflow) return set_error(-3, "wgrad: work list overflow");
  if (P <= 0) return 0;
  a.P = P;
  a.partials = scratch;
  if (a.ntile > 0) {
    static_assert(sizeof(WgradArgs) % 4 == 0 && sizeof(WgradArgs) <= 4096, "the lists travel as kernel arguments");
    WgradArgs* dev = (WgradArgs*)(scratch + wgrad_scratch_floats(P) - kWgradListFloats);
    const int nz = a.nz = wgrad_slices(P);
    static const int xcd = [] { const char* e = getenv("OBJNERF_WGRAD_XCD"); return e ? atoi(e) : 0; }();
    a.xcd = xcd;
    hipLaunchKernelGGL(wgrad_list_kernel, dim3(1), dim3(256), 0, s, a, dev);
    if (a.nfull > 0) hipLaunchKernelGGL(wgrad_units_kernel<false>, dim3(a.nfull * nz), dim3(256), 0, s, (const WgradArgs*)dev);
    if (a.ntile > a.nfull) hipLaunchKernelGGL(wgrad_units_kernel<true>, dim3((a.ntile - a.nfull) * nz), dim3(256), 0, s, (const WgradArgs*)dev);
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
