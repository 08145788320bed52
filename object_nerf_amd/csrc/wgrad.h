// wgrad.h -- all weight-gradient products of a model's backward in one grouped, deterministic pass.
//
// dW_l = dY_l^T X_l for the 20 Linear layers of an ObjectNeRF (28 products: the torch.cat inputs are column blocks): each
// product has only 1-6 output tiles of 128 x 128 but contracts over 10^5..10^6 sample points.  Round 2 launched one
// split-K GEMM per product (~70 launches per step incl. the ragged-tile launches) whose workgroups added their tiles with
// fp32 atomics: sums in arrival order, not reproducible run to run.  Here:
//   * ONE list of output tiles over all products; work unit = (tile, slice of ~4096 points), numbered so that the
//     workgroups resident together contract the same points for the neighbouring tiles of one product (operand panels
//     come from HBM once and are re-read from L2);
//   * every unit leaves its partial tile in its own scratch slot and wgrad_fixup_kernel adds a tile's slices in ascending
//     order: bit-reproducible gradients (tests/test_gpu_train.py), no atomics;
//   * two launches (full tiles, ragged tiles) + the fix-up per backward pass instead of ~35;
//   * the 1- and 3-row head products (sigma, rgb) are not MFMA work (a 128-row tile would be 97-99 % empty):
//     heads_wgrad_kernel reduces them on the VALU, also through ordered partial sums.
// The MFMA loop and the operand staging are gemm.h's (both operands row-contiguous: A' = dY^T, B' = X).
#pragma once
#include "gemm.h"

namespace objnerf {

struct WgProduct {             // dW (M x N, leading dimension ldc) += dY^T X
  const float* A; long lda;    // dY (P x .), A'[m][k] = A[k * lda + m]
  const float* B; long ldb;    // X  (P x .), B'[k][n] = B[k * ldb + n]
  float* C; long ldc;
  float* rowsum;               // bias gradient db[m] += sum_k dY[k][m] (taken by the tiles of the first tile column) or null
  int M, N;
  // optional, with rowsum, products without 256 x 256 tiles (round 5; full or ragged first column tile): the same column sums kept PER 16 POINTS,
  // segsum[(k / 16) * seg_ld + m] = sum of dY[k .. k + 15][m] -- the gradient w.r.t. anything that is constant along a ray
  // (the object code, the direction embedding: train.hip "per-ray terms") is a function of these sums, not of the points
  float* segsum; long seg_ld;
};
struct WgTile {                // one output tile: rows [128 by, +128), columns [128 bx, +128) of product `prod`
  unsigned char prod, by, bx;
  unsigned char ncol;          // 4: full tile (2 x 2 waves of 2 x 2 MFMA tiles); 1..3: ragged last column tile with that many
                               // live 32-column sub-tiles (4 x 1 waves, one row sub-tile each): ncol / 4 of a full tile's work
};
constexpr int kWgradMaxSlices = 256;                   // k slices per tile (fewer and longer ones beyond that)
constexpr int kWgradSliceIters = 64;                  // k iterations (of 32 points) per slice
constexpr int kWgradMaxProducts = 32;
constexpr int kWgradMaxTiles = 96;                    // capacity of the tile list (kernel argument)
// tiles the partial-tile scratch is sized for: the model's 28 products make 63 tiles of 128 x 128 with both branches in voxel
// mode (47 without the object branch, fewer in plain-PE mode) -- round 3 sized the slot area for all 96 list entries (1.5 GB at
// the reference batch); a pass with more tiles than this is refused by WgradBatch::launch
constexpr int kWgradSlotTiles = 64;
constexpr long kWgradSlotFloats = 128 * 128 + 128;    // a partial tile + its partial row sums
struct WgradArgs {             // passed by value (2.5 KB of kernel arguments: no device-side list to copy)
  WgProduct prod[kWgradMaxProducts];
  WgTile tile[kWgradMaxTiles];
  int nprod, ntile, nfull;     // tile[0 .. nfull) are full tiles, tile[nfull .. ntile) ragged ones
  int nbig;                    // tile[0 .. nbig) (a multiple of 4): the quadrants (0,0) (0,1) (1,0) (1,1) of 256 x 256 tiles, one
                               // workgroup each (wgrad_big_kernel); wgrad_units_kernel<false> takes tile[nbig .. nfull)
  int nz;                      // k slices per tile (and the slot stride of every tile)
  int nzb;                     // k slices per 256 x 256 tile, <= nz (round 6): as many as make ONE round of workgroups, see launch()
  int nzu, nzr;                // ... per 128 x 128 tile (two workgroups per CU) and per ragged tile (three); all <= nz
  int xcd;                     // XCD-aware unit map (tuning switch OBJNERF_WGRAD_XCD)
  long P;
  float* partials;             // slot (tile t, slice z) at (t * slices + z) * kWgradSlotFloats
};

struct HeadItem {              // dW (no x ni) += dY^T (no x P) X (P x ni), db (no) += column sums of dY; no <= 3, ni <= 256
  const float* dY; const float* X; float* dW; float* db;
  long ldx, ldw;
  int no, ni;
};
constexpr int kHeadChunk = 1024;                      // points per workgroup of heads_wgrad_kernel
constexpr int kHeadSlotFloats = 4 * 256;              // rows 0..2: partial dW[o][:], row 3: partial db[0..2]
constexpr int kMaxHeads = 4;
struct HeadArgs {
  HeadItem h[kMaxHeads];
  int nheads, nchunks;
  long P;
  float* partials;             // kMaxHeads x nchunks slots of kHeadSlotFloats
};

// floats of scratch the grouped weight-gradient pass needs for P points (appended to the dgrad scratch, train.hip)
constexpr long kWgradListFloats = 1024;               // device copy of the WgradArgs lists (<= 4 KB), at the end
// number of k slices of every tile (host): kWgradSliceIters k iterations each (OBJNERF_WGRAD_KITERS overrides: tuning),
// at most kWgradMaxSlices
int wgrad_slices(long P);
inline long wgrad_slot_floats(long P) { return (long)kWgradSlotTiles * wgrad_slices(P) * kWgradSlotFloats; }
inline long wgrad_scratch_floats(long P) {
  return wgrad_slot_floats(P) + (long)kMaxHeads * ((P + kHeadChunk - 1) / kHeadChunk) * kHeadSlotFloats + kWgradListFloats;
}

// host side: collects the products of one backward pass, then three launches on the stream
struct WgradBatch {
  WgradArgs a;
  HeadArgs h;
  bool overflow = false;
  WgradBatch() { a.nprod = a.ntile = a.nfull = a.nbig = 0; h.nheads = 0; }
  void add(const float* dY, long lddy, const float* X, long ldx, int out, int in, float* dW, long ldw, float* db,
           float* segsum = nullptr, long seg_ld = 0);
  void add_head(const float* dY, int no, const float* X, long ldx, int ni, float* dW, long ldw, float* db);
  int launch(long P, float* scratch, hipStream_t s, int slices = 0);
};

}  // namespace objnerf
