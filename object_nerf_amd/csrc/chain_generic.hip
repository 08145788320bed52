// chain_generic.hip -- runs of plain hidden layers of ANY architecture in one persistent kernel (round 6; SURVEY.md §8 rows a5 / a6 for
// config.model shapes other than the shipped default, models/nerf_model.py:41-58, 77-95: `nn.Sequential(nn.Linear(W, W), LeakyReLU)`
// for every layer that is neither the first nor in `skips`).
//
// The layer-wise path (generic.hip) evaluates such a network GEMM by GEMM: every W -> W layer reads its input rows from memory and
// writes its output rows back, at 0.54-0.63 of the fp32-MFMA peak on these shapes (profiles/r05_gemm_variants.txt).  A run of
// consecutive plain layers needs none of that: this kernel is the fused kernel's machine (mlp_kernel.h) with the layer list as
// a run-time argument -- 256 workgroups x 4 waves, a wave owns 32 points, the rows are read ONCE into the D register layout, every
// layer's weights stream through the 2-slot LDS ring as 32 KiB chunks (packed per call by chain_pack_kernel below: the same
// [4-k-step group][out tile][lane][k-step] chunk layout as the default architecture's stream, any W = 32 NT from 96 to 256;
// with one or two out tiles the compiler rejects layer_mac's DMA schedule, and such layers are a few microseconds of GEMM anyway), the D tile of
// one layer is the B operand of the next, and only the last layer's output is written.  A single plain layer takes the same kernel
// (L = 1): one read, one write, weights from LDS instead of a second operand panel per output tile.
#include <string.h>
#include "mlp_kernel.h"
#include "host_api.h"

namespace objnerf {

constexpr int kChainAuxFloats = 256;                       // per layer: NT x [half][16] bias values in D-register order
OBJ_HD constexpr int chain_ks(int nt) { return 16 * nt; }  // k-steps of a (32 nt) -> (32 nt) layer
OBJ_HD constexpr int chain_cpl(int nt) {                   // chunks per layer
  const int kg = chunk_ksteps(nt);
  return (chain_ks(nt) + kg - 1) / kg;
}

struct ChainPack {
  const float* W[kChainMaxLayers];      // nn.Linear weights, (out, in) row-major, W x W
  const float* b[kChainMaxLayers];
  int L, nt;
};
// stream position -> weight element, by the layout arithmetic load_group / layer_mac read it with (mlp_kernel.h; api.hip's
// objnerf_pack_index does the same for the default architecture on the host)
__global__ void __launch_bounds__(256) chain_pack_kernel(const ChainPack pk, float* __restrict__ blob, float* __restrict__ aux) {
  const int nt = pk.nt, W = 32 * nt, kg = chunk_ksteps(nt), ks_n = chain_ks(nt);
  const long per_layer = (long)chain_cpl(nt) * kChunkFloats, total = per_layer * pk.L;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int l = (int)(i / per_layer);
    const long r = i - (long)l * per_layer;
    const int chunk = (int)(r / kChunkFloats), e = (int)(r % kChunkFloats);
    const int slot = e >> 8, lane = (e >> 2) & 63, j = e & 3;
    const int g4 = slot / nt, m = slot % nt;
    const int ks = chunk * kg + 4 * g4 + j;
    float v = 0.f;                                         // slots past the chunk's k-steps / the layer's last k-step: padding
    if (4 * g4 < kg && ks < ks_n) v = pk.W[l][(long)(32 * m + (lane & 31)) * W + hid_feat(ks, lane >> 5)];
    blob[i] = v;
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)pk.L * nt * 32; i += (long)gridDim.x * 256) {
    const int l = (int)(i / (nt * 32)), e = (int)(i % (nt * 32));
    const int m = e >> 5, half = (e >> 4) & 1, r = e & 15;
    aux[l * kChainAuxFloats + e] = pk.b[l][32 * m + (r & 3) + 8 * (r >> 2) + 4 * half];
  }
}

struct ChainArgs {
  const float* blob; const float* aux;
  const float* X; long ldx;
  float* Y; long ldy;
  long P;
  int L, act_last;      // act_last: LeakyReLU after the last layer too (0: a layer without activation, e.g. xyz_encoding_final)
};

template <int NT>
__global__ void __launch_bounds__(256, 1) chain_kernel(const ChainArgs a, const long ntiles) {
  constexpr int kCB = kChunkBytes;
  __shared__ __attribute__((aligned(16))) char ring_mem[kRingSlots * kCB + kChainMaxLayers * kChainAuxFloats * 4 + kStageBytes];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, wave = tid >> 6;
  WeightStreamT<kCB> st;
  const int n_layers = __builtin_amdgcn_readfirstlane(a.L);
  st.init((const char*)a.blob, n_layers * chain_cpl(NT), (lds_char*)ring_mem, tid);
  float* aux_lds = (float*)(ring_mem + kRingSlots * kCB);
  for (int i = tid; i < n_layers * kChainAuxFloats; i += 256) aux_lds[i] = a.aux[i];
  __syncthreads();
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const Stage sg{(float*)(ring_mem + kRingSlots * kCB + kChainMaxLayers * kChainAuxFloats * 4) + wave * kStageFloats,
                   tile * 128 + wave * 32, a.P, lane};
    long p = sg.p0 + (lane & 31);
    if (p >= a.P) p = a.P - 1;                             // rows past the end repeat the last one; they are never stored
    f32x16 acc[NT], h[NT];
    // this lane's half of its point's row in the D layout: features 32 t + 8 g + 4 half .. + 3 are registers 4 g .. 4 g + 3 of tile t
    const float* xr = a.X + p * a.ldx + 4 * half;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = *(const f32x4u*)(xr + 32 * t + 8 * g);
        h[t][4 * g] = v[0]; h[t][4 * g + 1] = v[1]; h[t][4 * g + 2] = v[2]; h[t][4 * g + 3] = v[3];
      }
#pragma unroll 1
    for (int l = 0; l < n_layers; ++l) {
      const float* b = aux_lds + l * kChainAuxFloats + half * 16;
#pragma unroll
      for (int m = 0; m < NT; ++m) acc[m] = *(const f32x16*)(b + m * 32);
      HidSrc<NT> s{h};
      layer_mac<NT, chain_ks(NT), HidSrc<NT>>(acc, st, s);
      if (l + 1 < n_layers || a.act_last) finish<NT, true>(acc, h);     // (uniform)
      else finish<NT, false>(acc, h);
    }
    save_tiles<NT>(h, a.Y, a.ldy, sg);
  }
}

int64_t chain_scratch_floats(int width, int layers) {
  if (width < kChainMinWidth || width > 256 || (width & 31) || layers < 1) return 0;
  const int nt = width / 32;
  return (int64_t)layers * ((int64_t)chain_cpl(nt) * kChunkFloats + kChainAuxFloats);
}

// y = act(... act(x W_0^T + b_0) ... W_{L-1}^T + b_{L-1}) for L <= kChainMaxLayers layers of width 32 nt; x / y row-major with
// leading dimensions ldx / ldy (y may be x: a wave reads its 32 rows before it writes them).  scratch: chain_scratch_floats(width, L).
int launch_chain(int width, int L, const float* const* Ws, const float* const* bs, const float* X, long ldx, float* Y, long ldy, long P,
                 int act_last, float* scratch, hipStream_t s) {
  if (width < kChainMinWidth || width > 256 || (width & 31) || L < 1 || L > kChainMaxLayers || !X || !Y || !scratch)
    return set_error(-1, "chain: bad arguments (width a multiple of 32 from 96 to 256, 1 .. 16 layers)");
  if (P <= 0) return 0;
  const int nt = width / 32;
  ChainPack pk;
  for (int l = 0; l < L; ++l) { pk.W[l] = Ws[l]; pk.b[l] = bs[l]; }
  pk.L = L; pk.nt = nt;
  float* blob = scratch;
  float* aux = scratch + (long)L * chain_cpl(nt) * kChunkFloats;
  const long total = (long)L * chain_cpl(nt) * kChunkFloats;
  hipLaunchKernelGGL(chain_pack_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, s, pk, blob, aux);
  const ChainArgs a{blob, aux, X, ldx, Y, ldy, P, L, act_last};
  const long ntiles = (P + 127) / 128;
  const dim3 grid(mlp_grid(ntiles));
  switch (nt) {
    case 3: hipLaunchKernelGGL(chain_kernel<3>, grid, dim3(256), 0, s, a, ntiles); break;
    case 4: hipLaunchKernelGGL(chain_kernel<4>, grid, dim3(256), 0, s, a, ntiles); break;
    case 5: hipLaunchKernelGGL(chain_kernel<5>, grid, dim3(256), 0, s, a, ntiles); break;
    case 6: hipLaunchKernelGGL(chain_kernel<6>, grid, dim3(256), 0, s, a, ntiles); break;
    case 7: hipLaunchKernelGGL(chain_kernel<7>, grid, dim3(256), 0, s, a, ntiles); break;
    default: hipLaunchKernelGGL(chain_kernel<8>, grid, dim3(256), 0, s, a, ntiles); break;
  }
  return check_launch("chain");
}


// ---------------------------------------------------------------------------------------------------------------------------------
// A whole branch of any architecture in one persistent kernel (round 6): layers 0 .. D-1 -- the first layer and the skip layers read
// their `torch.cat([emb_xyz (, obj_voxel, obj_code) (, h)])` input (nerf_model.py:100-105, 128-138) as 32- or 64-column BLOCKS of the
// embedding rows straight from memory, the plain layers chain in registers as above -- then the 1-row density head on the VALU and the
// activation-free `final` layer.  What is left to the GEMMs of generic.hip is the direction layer (W -> W / 2 on cat([final, dir]))
// and the 3-row colour head.
//
// A memory block is a 16-k-step layer_mac of its own: lane (point, half) holds columns 8 g + 4 half .. + 3 (g = 0..3) of the block --
// four 16-byte pieces of its point's row, the same feature <-> (k-step, half) map as hid_feat -- and the block's weights are one
// chunk of the stream (its first 16 k-steps; columns past the tensor's width are zero).  The next block's pieces are requested
// before the current block's MFMAs (two register sets).
constexpr int kBrMaxLayers = 40;       // D + 1 (final)
constexpr int kBrMaxBlocks = 24;       // 32-column blocks of the branch's input tensors
struct BrLayer {
  const float* W; const float* b;      // nn.Linear weight (out = width, in = ldw) and bias
  int ldw;                             // in_features
  int nblk;                            // > 0: the layer contracts the branch's input blocks (first layer, skip layers)
  int hid_col0;                        // >= 0: ... and the previous layer's output, whose columns start here in W; -1: no hidden input
  int flags;                           // 1: LeakyReLU, 2: the density head reads this layer's output
  int chunk0;                          // first chunk of the layer in the stream
  int blk0;                            // its input blocks are blk[blk0 .. blk0 + nblk)
  float* save;                         // training: the layer's output rows (P x width) are also written here (null: not kept)
};
struct BrBlock { const float* x; long ld; int wcol0, col0, ncols, pad; };     // columns [col0, col0 + ncols) of x = W columns wcol0 ..
struct BrArgs {
  BrLayer layer[kBrMaxLayers];
  BrBlock blk[kBrMaxBlocks];
  int nlayers, nblocks, nt, total_chunks;
  int width, has_dir;                            // the branch's real width (<= 32 nt: 32- and 64-wide branches run on three out tiles, zero-padded)
  // has_dir: layer[nlayers] is the direction layer -- cat([final, emb_dir]) -> width / 2, LeakyReLU (nerf_model.py:116-118, 147-149) --
  // on dir_tiles(nt) out tiles, followed by the 3-row colour head + sigmoid: the kernel then writes rgb instead of final's rows
  const float* wrgb; const float* brgb;          // (3, width / 2), (3)
  float* rgb;                                    // (P, 3)
  float* save_dirh;                              // training: the direction layer's output rows (P x width / 2, width / 2 a multiple of 32)
  const float* wsig; const float* bsig;          // density head: (1, width), (1)
  float* sigma;                                  // (P)
  float* Y; long ldy;                            // output rows of the LAST layer (null: not wanted, sigma_only)
  long P;
  float* blob; float* aux;                       // packed stream / biases + head, in the workspace
};
static_assert(sizeof(BrArgs) <= 3584, "the branch description travels as a kernel argument");
// k-steps per input block: 16 (32 columns) fill a chunk for 8 out tiles; branches of up to 4 out tiles take 32 (64 columns) -- half
// the chunk barriers per MFMA and chunks filled to 32 of their k-steps instead of 16 (the 128-wide object branch of the default
// shape spends 60 % of its MFMAs in block layers: default shape 66.5 -> 67.9 M ray-samples/s); a block that holds at most 32 columns
// runs its first 16 k-steps only
OBJ_HD constexpr int blk_ks(int nt) { return nt <= 4 ? 32 : 16; }
// out tiles of the direction layer: width / 2 <= 96 columns on three tiles (zero-padded), more on four
OBJ_HD constexpr int dir_tiles(int nt) { return nt <= 6 ? 3 : 4; }
// aux: per layer 256 bias floats (the direction layer's too) | density head: 256 weights + bias | colour head: 3 x 128 weights + 3 biases
constexpr int kBrAuxFloats(int layers) { return (layers + 1) * kChainAuxFloats + kChainAuxFloats + 4 + 3 * 128 + 4; }

// grid (blocks over a layer's stream elements, layer): packs layer blockIdx.y's chunks; block (0, 0) also parks the description in
// device memory for branch_kernel (indexing a by-value kernel argument with a run-time index would give that kernel a private copy
// of the struct in scratch memory)
__global__ void __launch_bounds__(256) branch_pack_kernel(const BrArgs a, BrArgs* __restrict__ parked) {
  const int l = blockIdx.y;
  const bool dirl = a.has_dir && l == a.nlayers;           // the direction layer: its own out-tile count and real width
  const int nt = dirl ? dir_tiles(a.nt) : a.nt, kg = chunk_ksteps(nt), ks_n = chain_ks(a.nt);
  const int width_out = dirl ? a.width / 2 : a.width;
  const BrLayer ly = a.layer[l];
  const int hid_chunks = (ks_n + kg - 1) / kg;
  const int nch = ly.nblk + (ly.hid_col0 >= 0 ? hid_chunks : 0);
  float* out = a.blob + (long)ly.chunk0 * kChunkFloats;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)nch * kChunkFloats; i += (long)gridDim.x * 256) {
    const int chunk = (int)(i / kChunkFloats), e = (int)(i % kChunkFloats);
    const int slot = e >> 8, lane = (e >> 2) & 63, j = e & 3;
    const int g4 = slot / nt, m = slot % nt, half = lane >> 5;
    const int orow = 32 * m + (lane & 31);                  // output feature; rows / hidden columns past the real width: zero
    const long row = (long)orow * ly.ldw;
    float v = 0.f;
    if (orow >= width_out) {
    } else if (chunk < ly.nblk) {                           // a memory block: k-steps 0 .. 15 of its own chunk
      const BrBlock b = a.blk[ly.blk0 + chunk];
      const int ks = 4 * g4 + j, col = 8 * (ks >> 2) + (ks & 3) + 4 * half;
      if (ks < blk_ks(a.nt) && col < b.ncols) v = ly.W[row + b.wcol0 + col];
    } else {
      const int ks = (chunk - ly.nblk) * kg + 4 * g4 + j;
      if (4 * g4 < kg && ks < ks_n && hid_feat(ks, half) < a.width) v = ly.W[row + ly.hid_col0 + hid_feat(ks, half)];
    }
    out[i] = v;
  }
  if (blockIdx.x == 0) {
    float* const sig_aux = a.aux + (a.nlayers + 1) * kChainAuxFloats;      // behind the layers' (and the direction layer's) biases
    float* const rgb_aux = sig_aux + kChainAuxFloats + 4;
    for (int e = threadIdx.x; e < nt * 32; e += 256) {
      const int m = e >> 5, half = (e >> 4) & 1, r = e & 15, f = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * half;
      a.aux[l * kChainAuxFloats + e] = f < width_out ? ly.b[f] : 0.f;
      if (l == 0) sig_aux[e] = f < a.width ? a.wsig[f] : 0.f;
      if (dirl)
        for (int c = 0; c < 3; ++c) rgb_aux[c * 128 + e] = f < width_out ? a.wrgb[(long)c * width_out + f] : 0.f;
    }
    if (l == 0 && threadIdx.x == 0) sig_aux[kChainAuxFloats] = a.bsig[0];
    if (dirl && threadIdx.x < 3) rgb_aux[3 * 128 + threadIdx.x] = a.brgb[threadIdx.x];
    if (l == 0) {
      const unsigned* src = (const unsigned*)&a;
      unsigned* dst = (unsigned*)parked;
      for (unsigned i = threadIdx.x; i < sizeof(BrArgs) / 4; i += 256) dst[i] = src[i];
    }
  }
}

template <int KS>
struct RegSrc {
  const float (&v)[KS];
  template <int I>
  __device__ __forceinline__ float get() { return v[I]; }
};

template <int NT>
__global__ void __launch_bounds__(256, 1) branch_kernel(const BrArgs* __restrict__ ap, const long ntiles) {
  constexpr int kCB = kChunkBytes;
  constexpr int kAuxBytes = kBrAuxFloats(kBrMaxLayers) * 4;
  __shared__ __attribute__((aligned(16))) char ring_mem[kRingSlots * kCB + kAuxBytes + kStageBytes];
  __shared__ int lay_nblk[kBrMaxLayers + 1], lay_hid[kBrMaxLayers + 1], lay_flags[kBrMaxLayers + 1], lay_blk0[kBrMaxLayers + 1];
  __shared__ BrBlock blks[kBrMaxBlocks];
  __shared__ float* lay_save[kBrMaxLayers + 1];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, wave = tid >> 6;
  const int nl = __builtin_amdgcn_readfirstlane(ap->nlayers), nb = __builtin_amdgcn_readfirstlane(ap->nblocks);
  const long P = ap->P;
  const int has_dir = __builtin_amdgcn_readfirstlane(ap->has_dir);
  for (int i = tid; i < nl + has_dir; i += 256) {
    lay_nblk[i] = ap->layer[i].nblk; lay_hid[i] = ap->layer[i].hid_col0; lay_flags[i] = ap->layer[i].flags; lay_blk0[i] = ap->layer[i].blk0;
    lay_save[i] = ap->layer[i].save;
  }
  float* const save_dirh = ap->save_dirh;
  for (int i = tid; i < nb; i += 256) blks[i] = ap->blk[i];
  float* aux_lds = (float*)(ring_mem + kRingSlots * kCB);
  const float* gaux = ap->aux;
  for (int i = tid; i < kBrAuxFloats(nl); i += 256) aux_lds[i] = gaux[i];
  float* const rgb_out = ap->rgb;
  WeightStreamT<kCB> st;
  st.init((const char*)ap->blob, __builtin_amdgcn_readfirstlane(ap->total_chunks), (lds_char*)ring_mem, tid);
  const int nt_real = __builtin_amdgcn_readfirstlane(ap->width) >> 5;
  float* const sigma_out = ap->sigma;
  float* const Y = ap->Y;
  const long ldy = ap->ldy;
  __syncthreads();
  const float* aux_sig = aux_lds + (nl + 1) * kChainAuxFloats;
  const float* aux_rgb = aux_sig + kChainAuxFloats + 4;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const Stage sg{(float*)(ring_mem + kRingSlots * kCB + kAuxBytes) + wave * kStageFloats, tile * 128 + wave * 32, P, lane};
    const long p_raw = sg.p0 + (lane & 31);
    const long p = p_raw < P ? p_raw : P - 1;              // rows past the end repeat the last one; nothing of them is stored
    f32x16 acc[NT], h[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) h[m][r] = 0.f;
    constexpr int BKS = blk_ks(NT);
    // this lane's BKS values of input block e: columns 8 g + 4 half .. + 3 of the block are k-steps 4 g .. 4 g + 3
    auto fetch = [&](int e, float (&v)[BKS]) __attribute__((always_inline)) {
      const float* x = blks[e].x + p * blks[e].ld + blks[e].col0 + 4 * half;
      const int ncols = blks[e].ncols;
#pragma unroll
      for (int g = 0; g < BKS / 4; ++g) {
        const int c = 8 * g + 4 * half;
        if (c + 4 <= ncols) {
          const f32x4 q = *(const f32x4u*)(x + 8 * g);
          v[4 * g] = q[0]; v[4 * g + 1] = q[1]; v[4 * g + 2] = q[2]; v[4 * g + 3] = q[3];
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[4 * g + i] = c + i < ncols ? x[8 * g + i] : 0.f;
        }
      }
    };
    // one block's product; a 64-column block that holds at most 32 columns (a tensor's tail, a narrow tensor) runs its first 16
    // k-steps only -- the rest of its chunk is zeros
    auto mac = [&](int e, const float (&v)[BKS]) __attribute__((always_inline)) {
      RegSrc<BKS> s{v};
      if (BKS == 32 && __builtin_amdgcn_readfirstlane(blks[e].ncols) <= 32) layer_mac<NT, 16, RegSrc<BKS>>(acc, st, s);
      else layer_mac<NT, BKS, RegSrc<BKS>>(acc, st, s);
    };
#pragma unroll 1
    for (int l = 0; l < nl; ++l) {
      const int nblk = __builtin_amdgcn_readfirstlane(lay_nblk[l]), hid = __builtin_amdgcn_readfirstlane(lay_hid[l]);
      const int flags = __builtin_amdgcn_readfirstlane(lay_flags[l]);
      const float* b = aux_lds + l * kChainAuxFloats + half * 16;
#pragma unroll
      for (int m = 0; m < NT; ++m) acc[m] = *(const f32x16*)(b + m * 32);
      if (nblk > 0) {                                      // (uniform) the layer's input blocks, two register sets in turn
        const int b0 = __builtin_amdgcn_readfirstlane(lay_blk0[l]);
        float r0[BKS], r1[BKS];
        int e = 0;
        fetch(b0, r0);
        while (true) {
          if (e + 1 < nblk) fetch(b0 + e + 1, r1);
          mac(b0 + e, r0);
          if (++e >= nblk) break;
          if (e + 1 < nblk) fetch(b0 + e + 1, r0);
          mac(b0 + e, r1);
          if (++e >= nblk) break;
        }
      }
      if (hid >= 0) {
        HidSrc<NT> s{h};
        layer_mac<NT, chain_ks(NT), HidSrc<NT>>(acc, st, s);
      }
      if (flags & 1) finish<NT, true>(acc, h);
      else finish<NT, false>(acc, h);
      if (float* sv = lay_save[l]) {                       // (uniform) training: the layer's output is kept for the backward
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (t < nt_real) save_tile<NT>(h, t, sv, (long)(nt_real * 32), sg);
      }
      if (flags & 2) {                                     // density head on this layer's output (nerf_model.py:111, 141)
        const float sgm = head_dot<NT>(h, aux_sig, half) + aux_sig[kChainAuxFloats];
        if (half == 0 && p_raw < P) sigma_out[p_raw] = sgm;
      }
    }
    if (has_dir) {                                         // (uniform) direction layer on the final layer's output + colour head
      constexpr int NTO = dir_tiles(NT);
      f32x16 acc2[NTO], hd[NTO];
      const float* b = aux_lds + nl * kChainAuxFloats + half * 16;
#pragma unroll
      for (int m = 0; m < NTO; ++m) acc2[m] = *(const f32x16*)(b + m * 32);
      const int nblk = __builtin_amdgcn_readfirstlane(lay_nblk[nl]), b0 = __builtin_amdgcn_readfirstlane(lay_blk0[nl]);
      for (int e = 0; e < nblk; ++e) {                     // the direction embedding: one block for Embedding(3, 4), more for wider ones
        float r0[BKS];
        fetch(b0 + e, r0);
        RegSrc<BKS> s{r0};
        if (BKS == 32 && __builtin_amdgcn_readfirstlane(blks[b0 + e].ncols) <= 32) layer_mac<NTO, 16, RegSrc<BKS>>(acc2, st, s);
        else layer_mac<NTO, BKS, RegSrc<BKS>>(acc2, st, s);
      }
      {
        HidSrc<NT> s{h};
        layer_mac<NTO, chain_ks(NT), HidSrc<NT>>(acc2, st, s);
      }
      finish<NTO, true>(acc2, hd);
      if (save_dirh) {                                     // (uniform; width / 2 is a multiple of 32 then)
#pragma unroll
        for (int t = 0; t < NTO; ++t)
          if (2 * t < nt_real) save_tile<NTO>(hd, t, save_dirh, (long)(nt_real * 16), sg);
      }
      float col[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) col[c] = sigmoidf(head_dot<NTO>(hd, aux_rgb + c * 128, half) + aux_rgb[3 * 128 + c]);
      if (half == 0 && p_raw < P) { rgb_out[p_raw * 3] = col[0]; rgb_out[p_raw * 3 + 1] = col[1]; rgb_out[p_raw * 3 + 2] = col[2]; }
    } else if (Y) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (t < nt_real) save_tile<NT>(h, t, Y, ldy, sg);      // (uniform: the padding tiles of a narrow branch have no columns)
    }
  }
}

// chunks of a branch's stream: every block layer (the first + the skip layers) one chunk per input block, every layer with a hidden
// input (all but the first; + final) chain_cpl chunks
static long branch_chunks(int width, int D, int nskips, int nblocks, bool with_final) {
  const int nt = width / 32;
  return (long)(1 + nskips) * nblocks + (long)(D - 1 + (with_final ? 1 : 0)) * chain_cpl(nt);
}
static int dir_hid_chunks(int nt) { const int kg = chunk_ksteps(dir_tiles(nt)); return (chain_ks(nt) + kg - 1) / kg; }
static int blocks_of(int cols, int bw) { return (cols + bw - 1) / bw; }     // bw = 2 blk_ks(nt) columns per block
static int branch_width(int width) { return width < kChainMinWidth ? kChainMinWidth : width; }       // the width the kernel runs at
int64_t branch_scratch_floats(int width, int D, int nskips, int in_a, int in_b, int in_c, int in_dir) {
  if (width < 32 || width > 256 || (width & 31) || D + 1 > kBrMaxLayers) return 0;
  width = branch_width(width);
  const int bw = 2 * blk_ks(width / 32);
  const int nb = blocks_of(in_a, bw) + (in_b > 0 ? blocks_of(in_b, bw) : 0) + (in_c > 0 ? blocks_of(in_c, bw) : 0);
  if (nb > kBrMaxBlocks) return 0;
  // (the direction layer's part whenever launch_branch could take it: an upper bound of what any call of this shape packs)
  const long chunks = branch_chunks(width, D, nskips, nb, true) + blocks_of(in_dir > 0 ? in_dir : 0, bw) + dir_hid_chunks(width / 32);
  return chunks * kChunkFloats + kBrAuxFloats(D + 1) + (int64_t)(sizeof(BrArgs) + 3) / 4 + 8;
}

// One branch: q = the branch's parameter pointers in objnerf_arch order (D layers' weight, bias; then final, dir, sigma, rgb),
// in[] = its input tensors (row-major, `c` columns each, concatenated in this order by the reference).  Writes sigma (P) and, unless
// sigma_only, the final layer's rows to fin (P x width).  Returns 1 when the shape is not one this kernel takes (the caller then
// runs its GEMMs), 0 on success, < 0 on error.
// emb_dir (P x in_dir) + rgb (P x 3) given: the direction layer and the colour head run in the kernel too and `fin` is not written.
int launch_branch(int width, int D, const int32_t* skips, int nskips, const float* const* q, const BranchInput* in, int nin, long P,
                  float* sigma, float* fin, bool sigma_only, const float* emb_dir, int in_dir, float* rgb, float* scratch,
                  hipStream_t s, float* const* saves, float* save_dirh) {
  if (width < 32 || width > 256 || (width & 31) || D + 1 > kBrMaxLayers || D < 1) return 1;
  const int width_real = width;
  width = branch_width(width);                  // 32- and 64-wide branches: three out tiles, the surplus zero (layer_mac does not
                                                // compile for one or two out tiles)
  BrArgs a;
  memset(&a, 0, sizeof(a));
  const int bw = 2 * blk_ks(width / 32);        // columns per input block
  int nb = 0, cin = 0;
  for (int i = 0; i < nin; ++i) {
    for (int c0 = 0; c0 < in[i].c; c0 += bw) {
      if (nb >= kBrMaxBlocks) return 1;
      a.blk[nb++] = BrBlock{in[i].x, (long)in[i].c, cin + c0, c0, in[i].c - c0 < bw ? in[i].c - c0 : bw, 0};
    }
    cin += in[i].c;
  }
  // (training keeps the direction layer's output: its rows are whole 32-column tiles only when width / 2 is a multiple of 32)
  const bool with_dir = !sigma_only && emb_dir && rgb && in_dir > 0 && (width_real & 1) == 0 && nb + blocks_of(in_dir, bw) <= kBrMaxBlocks &&
                        (!save_dirh || (width_real / 2) % 32 == 0);
  const int nt = width / 32;
  auto is_skip = [&](int l) { for (int i = 0; i < nskips; ++i) if (skips[i] == l) return true; return false; };
  int chunk = 0, nl = 0;
  for (int l = 0; l < D; ++l) {
    const bool blk_layer = l == 0 || is_skip(l);
    BrLayer& y = a.layer[nl++];
    y.W = q[2 * l]; y.b = q[2 * l + 1];
    y.nblk = blk_layer ? nb : 0;
    y.hid_col0 = l == 0 ? -1 : (blk_layer ? cin : 0);
    y.ldw = (blk_layer ? cin : 0) + (l == 0 ? 0 : width_real);
    y.flags = 1 | (l == D - 1 ? 2 : 0);
    y.chunk0 = chunk;
    y.blk0 = 0;
    y.save = saves ? saves[l] : nullptr;
    chunk += y.nblk + (y.hid_col0 >= 0 ? chain_cpl(nt) : 0);
  }
  const float* const* t = q + 2 * D;            // final, dir, sigma, rgb
  if (!sigma_only) {
    BrLayer& y = a.layer[nl++];
    y.W = t[0]; y.b = t[1]; y.nblk = 0; y.hid_col0 = 0; y.ldw = width_real; y.flags = 0; y.chunk0 = chunk; y.blk0 = 0;
    y.save = (saves && with_dir) ? fin : nullptr;          // (without the direction layer `fin` is the kernel's output anyway)
    chunk += chain_cpl(nt);
  }
  int nb_all = nb;
  if (with_dir) {                               // layer[nl]: cat([final, emb_dir]) -> width / 2 (its hidden columns lead the weight)
    BrLayer& y = a.layer[nl];
    y.W = t[2]; y.b = t[3]; y.blk0 = nb; y.nblk = blocks_of(in_dir, bw); y.hid_col0 = 0; y.ldw = width_real + in_dir; y.flags = 1; y.chunk0 = chunk;
    for (int c0 = 0; c0 < in_dir; c0 += bw)
      a.blk[nb_all++] = BrBlock{emb_dir, (long)in_dir, width_real + c0, c0, in_dir - c0 < bw ? in_dir - c0 : bw, 0};
    chunk += y.nblk + dir_hid_chunks(nt);
    a.has_dir = 1; a.wrgb = t[6]; a.brgb = t[7]; a.rgb = rgb; a.save_dirh = save_dirh;
    y.save = nullptr;
  }
  a.nlayers = nl; a.nblocks = nb_all; a.nt = nt; a.total_chunks = chunk;
  a.wsig = t[4]; a.bsig = t[5];
  a.width = width_real;
  a.sigma = sigma; a.Y = sigma_only ? nullptr : fin; a.ldy = width_real; a.P = P;
  a.blob = scratch;
  a.aux = scratch + (long)chunk * kChunkFloats;
  BrArgs* parked = (BrArgs*)(a.aux + kBrAuxFloats(nl) + ((4 - (kBrAuxFloats(nl) & 3)) & 3));
  if (P <= 0) return 0;
  const int max_ch = nb + chain_cpl(nt) + dir_hid_chunks(nt);
  hipLaunchKernelGGL(branch_pack_kernel, dim3((unsigned)((max_ch * kChunkFloats + 4095) / 4096), (unsigned)(nl + (with_dir ? 1 : 0))), dim3(256), 0, s,
                     a, parked);
  const long ntiles = (P + 127) / 128;
  const dim3 grid(mlp_grid(ntiles));
  switch (nt) {
    case 3: hipLaunchKernelGGL(branch_kernel<3>, grid, dim3(256), 0, s, (const BrArgs*)parked, ntiles); break;
    case 4: hipLaunchKernelGGL(branch_kernel<4>, grid, dim3(256), 0, s, (const BrArgs*)parked, ntiles); break;
    case 5: hipLaunchKernelGGL(branch_kernel<5>, grid, dim3(256), 0, s, (const BrArgs*)parked, ntiles); break;
    case 6: hipLaunchKernelGGL(branch_kernel<6>, grid, dim3(256), 0, s, (const BrArgs*)parked, ntiles); break;
    case 7: hipLaunchKernelGGL(branch_kernel<7>, grid, dim3(256), 0, s, (const BrArgs*)parked, ntiles); break;
    default: hipLaunchKernelGGL(branch_kernel<8>, grid, dim3(256), 0, s, (const BrArgs*)parked, ntiles); break;
  }
  const int rc = check_launch("branch");
  return rc ? rc : (with_dir ? 2 : 0);          // 2: sigma AND rgb are written; 0: sigma and fin
}

}  // namespace objnerf
