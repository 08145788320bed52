// mlp_bwd.hip -- training: the dgrad chain through the hidden layers of both MLP branches in ONE persistent kernel
// (what loss.backward() does through models/nerf_model.py:97-152 for the hidden activations).
//
//   d(pre-activation of layer l-1) = leaky'(A_{l-1}) . ( W_l[:, hidden block]^T  d(pre-activation of layer l) )
//
// Same machine as the forward (mlp_kernel.h): 256 workgroups x 4 waves, a wave owns 32 sample points, the transposed
// weight blocks stream through the 2-slot LDS ring (layout.h: "backward weight stream"), and the D tile of one layer
// -- the gradient, features x points in registers -- is the B operand of the next, so the chain never leaves the
// register file.  Per layer the kernel only reads the saved forward activation (for the LeakyReLU mask, fetched
// before the layer's MFMAs and consumed after them) and writes the gradient w.r.t. the layer's pre-activation output,
// which the weight-gradient GEMMs and the embedding-gradient GEMMs of train.hip consume afterwards.
// Heads: d(dir hidden) = t2 * W_rgb and "+ d_sigma * w_sigma" are 3- and 1-term VALU updates from the aux block.
#include "mlp_kernel.h"
#include "host_api.h"

namespace objnerf {

struct BwdArgs {
  const float* blob_bwd;
  const float* aux;
  long P;
  const float* act;     // forward workspace (SaveWs layout)
  float* dz;            // gradients w.r.t. pre-activation outputs, same layout
  const float* d_sigma; // (P)
  const float* t2;      // (P,3) gradient w.r.t. the rgb head's pre-sigmoid output
  const float* d_isigma;
  const float* t2i;
};

template <int NT>
__device__ __forceinline__ void load_tiles(f32x16 (&v)[NT], const float* mat, long ld, long p, int half) {
#ifdef OBJ_ABL_BWD_NOLOAD     // timing ablation only
  for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) v[t][r] = 1.f;
  return;
#endif
  const float* row = mat + p * ld + 4 * half;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 x = *(const f32x4*)(row + 32 * t + 8 * g);
      v[t][4 * g] = x[0]; v[t][4 * g + 1] = x[1]; v[t][4 * g + 2] = x[2]; v[t][4 * g + 3] = x[3];
    }
}
template <int NT>
__device__ __forceinline__ void zero_tiles(f32x16 (&acc)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}
// h = leaky'(act) . acc      (leaky_relu backward on sign(output) = sign(input))
template <int NT>
__device__ __forceinline__ void mask_tiles(const f32x16 (&acc)[NT], const f32x16 (&act)[NT], f32x16 (&h)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) h[t][r] = act[t][r] > 0.f ? acc[t][r] : 0.01f * acc[t][r];
}
// acc[t][r] += s * head_row[(t, half, r)]   (gradient of a 1-output head: outer product with its weight row)
template <int NT>
__device__ __forceinline__ void add_head(f32x16 (&acc)[NT], const float* w, int half, float s) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f32x16 wv = *(const f32x16*)(w + (t * 2 + half) * 16);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = fmaf(s, wv[r], acc[t][r]);
  }
}

template <bool DO_OBJ>
__global__ void __launch_bounds__(256, 1) mlp_bwd_kernel(const BwdArgs a, const long ntiles) {
  __shared__ __attribute__((aligned(16))) char ring_mem[kRingSlots * kChunkBytes + kAuxFloats * 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int wave = tid >> 6;
  WeightStream st;
  st.init((const char*)a.blob_bwd, DO_OBJ ? bwd_total_chunks() : bwd_scene_chunks(), (lds_char*)ring_mem, tid);
  float* aux_lds = (float*)(ring_mem + kRingSlots * kChunkBytes);
  for (int i = tid; i < kAuxFloats; i += 256) aux_lds[i] = a.aux[i];
  __syncthreads();
  const float* aux = aux_lds;
  const long P = a.P;
  const SaveWs act{const_cast<float*>(a.act), P};
  const SaveWs dz{a.dz, P};

  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long p_raw = tile * 128 + wave * 32 + (lane & 31);
    const bool valid = p_raw < P;
    const long p = valid ? p_raw : P - 1;

    // Stores of a layer's result and the fetch of the activation that masks the NEXT result are issued from the
    // after-barrier hook of the layer that consumes the former (layer_mac): they then have a chunk of MFMAs to land
    // instead of being drained by the very next barrier.
    // ---------------- scene branch ----------------
    {
      f32x16 acc[8], h[8], av[8];
      {
        // d(dir hidden) = t2 (P,3) * W_rgb (3,128), masked by the dir layer's LeakyReLU
        f32x16 hd[4], ad[4];
        load_tiles<4>(ad, act.sdirh(), 128, p, half);
        zero_tiles<4>(hd);
#pragma unroll
        for (int c = 0; c < 3; ++c) add_head<4>(hd, aux + kAuxSRgb + c * 4 * 32, half, a.t2[p * 3 + c]);
        mask_tiles<4>(hd, ad, hd);
        // BL_SD: -> d(xyz_encoding_final output), no activation there
        zero_tiles<8>(acc);
        {
          HidSrc<4> s{hd};
          layer_mac<8, bwd_ks(BL_SD)>(acc, st, s, [&]() __attribute__((always_inline)) {
            save_tiles<4>(hd, dz.sdirh(), 128, p, half, valid);
          });
        }
      }
      finish<8, false>(acc, h);
      // BL_SF: -> dA8, plus the density head's contribution, then layer 8's mask
      zero_tiles<8>(acc);
      {
        HidSrc<8> s{h};
        layer_mac<8, 128>(acc, st, s, [&]() __attribute__((always_inline)) {
          save_tiles<8>(h, dz.sfinal(), 256, p, half, valid);
          load_tiles<8>(av, act.A(8), 256, p, half);
        });
      }
      add_head<8>(acc, aux + kAuxSSig, half, a.d_sigma[p]);
      mask_tiles<8>(acc, av, h);
      // BL_S8 .. BL_S2 (BL_S5 streams the hidden block of the skip layer): dZ_l -> dZ_{l-1}
#pragma unroll 1
      for (int l = 8; l >= 2; --l) {
        zero_tiles<8>(acc);
        {
          HidSrc<8> s{h};
          layer_mac<8, 128>(acc, st, s, [&]() __attribute__((always_inline)) {
            save_tiles<8>(h, dz.A(l), 256, p, half, valid);
            load_tiles<8>(av, act.A(l - 1), 256, p, half);
          });
        }
        mask_tiles<8>(acc, av, h);
      }
      save_tiles<8>(h, dz.A(1), 256, p, half, valid);
    }

    // ---------------- object branch ----------------
    if constexpr (DO_OBJ) {
      f32x16 acc[4], h[4], av[4];
      {
        f32x16 hd[2], ad[2];
        load_tiles<2>(ad, act.odirh(), 64, p, half);
        zero_tiles<2>(hd);
#pragma unroll
        for (int c = 0; c < 3; ++c) add_head<2>(hd, aux + kAuxORgb + c * 2 * 32, half, a.t2i[p * 3 + c]);
        mask_tiles<2>(hd, ad, hd);
        zero_tiles<4>(acc);
        {
          HidSrc<2> s{hd};
          layer_mac<4, bwd_ks(BL_OD)>(acc, st, s, [&]() __attribute__((always_inline)) {
            save_tiles<2>(hd, dz.odirh(), 64, p, half, valid);
          });
        }
      }
      finish<4, false>(acc, h);
      zero_tiles<4>(acc);
      {
        HidSrc<4> s{h};
        layer_mac<4, 64>(acc, st, s, [&]() __attribute__((always_inline)) {
          save_tiles<4>(h, dz.ofinal(), 128, p, half, valid);
          load_tiles<4>(av, act.B(4), 128, p, half);
        });
      }
      add_head<4>(acc, aux + kAuxOSig, half, a.d_isigma[p]);
      mask_tiles<4>(acc, av, h);
#pragma unroll 1
      for (int l = 4; l >= 2; --l) {
        zero_tiles<4>(acc);
        {
          HidSrc<4> s{h};
          layer_mac<4, 64>(acc, st, s, [&]() __attribute__((always_inline)) {
            save_tiles<4>(h, dz.B(l), 128, p, half, valid);
            load_tiles<4>(av, act.B(l - 1), 128, p, half);
          });
        }
        mask_tiles<4>(acc, av, h);
      }
      save_tiles<4>(h, dz.B(1), 128, p, half, valid);
    }
  }
}

int launch_mlp_bwd(const float* blob_bwd, const float* aux, long P, const float* act, float* dz, const float* d_sigma,
                   const float* t2, const float* d_isigma, const float* t2i, bool do_object, hipStream_t s) {
  static_assert(bwd_scene_chunks() == 68 || kChunkTiles != 128, "backward stream layout");
  const BwdArgs a{blob_bwd, aux, P, act, dz, d_sigma, t2, d_isigma, t2i};
  const long ntiles = (P + 127) / 128;
  const unsigned grid = mlp_grid(ntiles);
  if (do_object) hipLaunchKernelGGL((mlp_bwd_kernel<true>), dim3(grid), dim3(256), 0, s, a, ntiles);
  else hipLaunchKernelGGL((mlp_bwd_kernel<false>), dim3(grid), dim3(256), 0, s, a, ntiles);
  return check_launch("mlp_train_backward(fused)");
}

}  // namespace objnerf
