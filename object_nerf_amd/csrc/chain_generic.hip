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

}  // namespace objnerf
