// mlp_bwd.hip -- training: the dgrad chain through the hidden layers of both MLP branches in ONE persistent kernel
// (what loss.backward() does through models/nerf_model.py:97-152 for the hidden activations).
//
//   d(pre-activation of layer l-1) = leaky'(A_{l-1}) . ( W_l[:, hidden block]^T  d(pre-activation of layer l) )
//
// MASKS = true (round 5): leaky' comes from the sign masks the training forward packed in this very register layout
// (mlp_kernel.h "LeakyReLU masks": one 16-byte load per lane and layer) instead of from the re-read activation matrices
// (MASKS = false: 1.1 GB per launch through the wave's LDS patch; kept for a forward that ran layer by layer on the GEMMs
// and left no masks).  Same bits either way: identical gradients.
//
// Same machine as the forward (mlp_kernel.h): 256 workgroups x 4 waves, a wave owns 32 sample points, the transposed
// weight blocks stream through the 2-slot LDS ring (layout.h: "backward weight stream"), and the D tile of one layer
// -- the gradient, features x points in registers -- is the B operand of the next, so the chain never leaves the
// register file.  Per layer the kernel only reads the saved forward activation (for the LeakyReLU mask, fetched
// under the layer's MFMAs and consumed after them) and writes the gradient w.r.t. the layer's pre-activation output,
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
  const unsigned* masks; // MASKS: the forward's sign masks (mlp_kernel.h layout)
  float* d_emb;          // DX: gradient w.r.t. the embedding rows (P, ld_emb), only the kScnVoxPE voxel-feature columns are written
  float* d_ov;           // DX (object branch): gradient w.r.t. the object voxel embedding (P, kObjVoxPE)
  long ld_emb;
};
#ifndef OBJ_BWD_SPREAD_SAVE
#define OBJ_BWD_SPREAD_SAVE 1   // MASKS: a layer's input-gradient tiles are stored one per MFMA group (the registers the raw
                                // activation tiles needed are free) instead of as one burst behind the layer's first barrier
#endif

// Saved activations come back the way save_tiles wrote them: 8 consecutive lanes read one 128-byte line of a point.
// Only their signs are needed (LeakyReLU mask), and registers are tight (gradient in + gradient out = 256 of 512),
// so a layer fetches them four tiles at a time behind one chunk barrier and, behind the next one, turns them into
// the D layout through the wave's LDS patch and keeps one bit per value.
struct RawTiles { f32x4 v[4][4]; };          // 4 tiles x 4 row groups, as fetched (lane = (row q, piece k))
template <int N>
__device__ __forceinline__ void fetch_tiles(RawTiles& raw, const float* mat, long ld, int t0, const Stage& sg) {
  const int q = sg.lane >> 3, k = sg.lane & 7;
#pragma unroll
  for (int t = 0; t < N; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long row = sg.p0 + 8 * i + q;
#if OBJ_NT_ACT
      raw.v[t][i] = __builtin_nontemporal_load((const f32x4*)(mat + (row < sg.P ? row : sg.P - 1) * ld + 32 * (t0 + t) + 4 * k));
#else
      raw.v[t][i] = *(const f32x4*)(mat + (row < sg.P ? row : sg.P - 1) * ld + 32 * (t0 + t) + 4 * k);
#endif
    }
}
template <int NT, int N, int T0>
__device__ __forceinline__ void sign_bits(const RawTiles& raw, MaskBits<NT>& bits, const Stage& sg) {
  const int pt = sg.lane & 31, half = sg.lane >> 5, q = sg.lane >> 3, k = sg.lane & 7;
#pragma unroll
  for (int t = 0; t < N; ++t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *(f32x4*)(sg.buf + (8 * i + q) * kStageLd + 4 * k) = raw.v[t][i];
    unsigned m = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 act = *(const f32x4*)(sg.buf + pt * kStageLd + 8 * g + 4 * half);
#pragma unroll
      for (int j = 0; j < 4; ++j) m |= (act[j] > 0.f ? 1u : 0u) << (4 * g + j);
    }
    if (((T0 + t) & 1) == 0) bits.w[(T0 + t) / 2] = m;       // t is unrolled: resolved at compile time
    else bits.w[(T0 + t) / 2] |= m << 16;
  }
}
// h = leaky'(act) . acc      (leaky_relu backward on sign(output) = sign(input))
#ifndef OBJ_BWD_CARRY_MASK
#define OBJ_BWD_CARRY_MASK 1
#endif
template <int NT>
__device__ __forceinline__ void mask_tiles(const f32x16 (&acc)[NT], const MaskBits<NT>& bits, f32x16 (&h)[NT]) {
#if OBJ_BWD_CARRY_MASK
  // The mask word is consumed from its top bit down: `m + m` shifts it left and leaves the bit that fell out in VCC, which the
  // select reads directly -- one VALU instruction where "extract bit, compare" took two (this kernel carries 1.17 VALU
  // instructions per MFMA and fp32 MFMA shares the SIMD's ALUs with them, profiles/r04_train_pmc.md).  Same selects, same
  // products: the results are bit-equal.
#pragma unroll
  for (int wd = 0; wd < NT / 2; ++wd) {
    unsigned m = bits.w[wd];
#pragma unroll
    for (int tt = 1; tt >= 0; --tt) {
      const int t = 2 * wd + tt;
#pragma unroll
      for (int r = 15; r >= 0; r -= 2) {
        const f32x2 v = {acc[t][r - 1], acc[t][r]};
        const f32x2 sl = v * 0.01f;                  // v_pk_mul_f32
        float o1, o0;
        asm volatile("v_add_co_u32 %0, vcc, %0, %0\n\tv_cndmask_b32 %1, %2, %3, vcc"
                     : "+v"(m), "=v"(o1) : "v"(sl[1]), "v"(v[1]) : "vcc");
        asm volatile("v_add_co_u32 %0, vcc, %0, %0\n\tv_cndmask_b32 %1, %2, %3, vcc"
                     : "+v"(m), "=v"(o0) : "v"(sl[0]), "v"(v[0]) : "vcc");
        h[t][r] = o1;
        h[t][r - 1] = o0;
      }
    }
  }
#else
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      h[t][r] = ((bits.w[t / 2] >> (16 * (t & 1) + r)) & 1u) ? acc[t][r] : 0.01f * acc[t][r];
#endif
}
// after-barrier hook of one backward layer: chunk 0 stores the layer's input gradient (the previous layer's result)
// and fetches the first four activation tiles of the mask of ITS result; chunks 1, 2 reduce them to bits / fetch the rest
template <int NTIN, int NTOUT, bool MASKS>
struct BwdHook {
  const f32x16 (&hin)[NTIN];
  float* save_mat; long save_ld;
  const float* act_mat; long act_ld;        // MASKS = false: the saved output whose signs mask the layer's result (nullptr: not masked)
  const unsigned* mrow;                     // MASKS = true: this lane's 16 bytes of that mask (nullptr: not masked)
  RawTiles& raw;
  MaskBits<NTOUT>& bits;
  const Stage& sg;
  // MASKS = false: per-group spreading of the stores, as the forward's SaveHook does, overflows the register budget (gradient
  // in + gradient out + raw activation tiles leave no room for the store temporaries mid-layer)
  static constexpr int min_groups = (MASKS && OBJ_BWD_SPREAD_SAVE) ? NTIN : 0;
  template <int GI>
  __device__ __forceinline__ void group() const {
    if constexpr (MASKS && OBJ_BWD_SPREAD_SAVE && GI < NTIN) save_tile<NTIN>(hin, GI, save_mat, save_ld, sg);
  }
  template <int C>
  __device__ __forceinline__ void operator()(std::integral_constant<int, C>) const {
    if constexpr (MASKS) {
      if constexpr (C == 0) {
        if constexpr (!OBJ_BWD_SPREAD_SAVE) save_tiles<NTIN>(hin, save_mat, save_ld, sg);
        if (mrow) load_masks<NTOUT>(bits, mrow);
      }
    } else if constexpr (C == 0) {
      save_tiles<NTIN>(hin, save_mat, save_ld, sg);
      if (act_mat) fetch_tiles<4>(raw, act_mat, act_ld, 0, sg);
    } else if constexpr (C == 1) {
      if (act_mat) {
        sign_bits<NTOUT, 4, 0>(raw, bits, sg);
        if constexpr (NTOUT == 8) fetch_tiles<4>(raw, act_mat, act_ld, 4, sg);
      }
    } else if constexpr (C == 2 && NTOUT == 8) {
      if (act_mat) sign_bits<NTOUT, 4, 4>(raw, bits, sg);
    }
  }
};
// A tile of the embedding gradient to its rows of a (P, ld) matrix whose row stride is NOT a multiple of 16 bytes (ld = 271) and
// whose last tile is only partly there (`cols` valid columns in all): through the wave's LDS patch like save_tile (8 lanes = one
// point's 128 bytes), unaligned 16-byte stores, pieces beyond `cols` dropped.
template <int NT>
__device__ __forceinline__ void save_tile_cols(const f32x16 (&h)[NT], int t, float* mat, long ld, int cols, const Stage& sg) {
  const int pt = sg.lane & 31, half = sg.lane >> 5, q = sg.lane >> 3, k = sg.lane & 7;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 v = {h[t][4 * g], h[t][4 * g + 1], h[t][4 * g + 2], h[t][4 * g + 3]};
    *(f32x4*)(sg.buf + pt * kStageLd + 8 * g + 4 * half) = v;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + q;
    const f32x4 v = *(const f32x4*)(sg.buf + row * kStageLd + 4 * k);
    if (sg.p0 + row < sg.P && 32 * t + 4 * k + 4 <= cols) *(f32x4u*)(mat + (sg.p0 + row) * ld + 32 * t + 4 * k) = v;
  }
}
template <int NT>
__device__ __forceinline__ void zero_tiles(f32x16 (&acc)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
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

// DX (voxel mode, MASKS): the gradients w.r.t. the voxel-feature columns of the embeddings are formed HERE -- while dZ5 / dZ1 / dB3 /
// dB1 are in registers they are also contracted with those layers' embedding-column blocks (BL_X* of the stream) into accx / accxo,
// which stay in registers to the end of the tile and are written once (312 floats per point).  Rounds 2-5 ran two segmented GEMMs
// over the four gradient matrices afterwards (1.5 ms of an 18.6 ms step at 0.61 of peak, co-bound by re-reading what this kernel had
// just written).
template <bool DO_OBJ, bool MASKS, bool DX>
__global__ void __launch_bounds__(256, 1) mlp_bwd_kernel(const BwdArgs a, const long ntiles) {
  static_assert(!DX || MASKS, "the embedding-gradient fold rides on the mask-fed chain");
  constexpr int kCB = kChunkBytes;
  __shared__ __attribute__((aligned(16))) char ring_mem[kRingSlots * kCB + kAuxFloats * 4 + kStageBytes];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int wave = tid >> 6;
  WeightStreamT<kCB> st;
  st.init((const char*)a.blob_bwd, DO_OBJ ? bwd_total_chunks(DX) : bwd_scene_chunks(DX), (lds_char*)ring_mem, tid);
  float* aux_lds = (float*)(ring_mem + kRingSlots * kCB);
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

    const Stage sg_scene{(float*)(ring_mem + kRingSlots * kCB + kAuxFloats * 4) + wave * kStageFloats,
                         tile * 128 + wave * 32, P, lane};
    // Stores of a layer's result and the fetch of the activation that masks the NEXT result are issued from the
    // after-barrier hooks of the layer that consumes the former (layer_mac): they then have a chunk of MFMAs to land
    // instead of being drained by the very next barrier.
    RawTiles raw;
    // this lane's 16 bytes of mask group grp of the wave's 32 points
    auto mrow_of = [&](int grp) __attribute__((always_inline)) {
      return MASKS ? a.masks + (sg_scene.p0 >> 5) * kMaskDwordsPerWave + ((long)grp * 64 + lane) * 4 : nullptr;
    };
    f32x16 accx[DX ? 7 : 1];             // d(scene voxel embedding): 208 columns (7 tiles), live from X5 to the end of the tile
    // ---------------- scene branch ----------------
    {
      const Stage& sg = sg_scene;
      f32x16 acc[8], h[8];
      MaskBits<8> bits;
      {
        // d(dir hidden) = t2 (P,3) * W_rgb (3,128), masked by the dir layer's LeakyReLU
        f32x16 hd[4];
        MaskBits<4> bd;
        if constexpr (MASKS) load_masks<4>(bd, mrow_of(kMaskGrpSD));
        else fetch_tiles<4>(raw, act.sdirh(), 128, 0, sg);
        zero_tiles<4>(hd);
#pragma unroll
        for (int c = 0; c < 3; ++c) add_head<4>(hd, aux + kAuxSRgb + c * 4 * 32, half, a.t2[p * 3 + c]);
        if constexpr (!MASKS) sign_bits<4, 4, 0>(raw, bd, sg);
        mask_tiles<4>(hd, bd, hd);
        // BL_SD: -> d(xyz_encoding_final output), no activation there
        {
          HidSrc<4> s{hd};
          layer_mac<8, bwd_ks(BL_SD), HidSrc<4>, BwdHook<4, 8, MASKS>, true>(acc, st, s, {hd, dz.sdirh(), 128, nullptr, 0, nullptr, raw, bits, sg});
        }
      }
      finish<8, false>(acc, h);
      // BL_SF: -> dA8, plus the density head's contribution, then layer 8's mask
      {
        HidSrc<8> s{h};
        layer_mac<8, 128, HidSrc<8>, BwdHook<8, 8, MASKS>, true>(acc, st, s, {h, dz.sfinal(), 256, act.A(8), 256, mrow_of(kMaskGrpA + 7), raw, bits, sg});
      }
      add_head<8>(acc, aux + kAuxSSig, half, a.d_sigma[p]);
      mask_tiles<8>(acc, bits, h);
      // BL_S8 .. BL_S2 (BL_S5 streams the hidden block of the skip layer): dZ_l -> dZ_{l-1}
#pragma unroll 1
      for (int l = 8; l >= 2; --l) {
        if constexpr (DX) {
          if (l == 5) {                 // (uniform) BL_X5: h = dZ5 also meets the skip layer's embedding columns
            HidSrc<8> s{h};
            layer_mac<7, bwd_ks(BL_X5), HidSrc<8>, NoHook, true>(accx, st, s);
          }
        }
        {
          HidSrc<8> s{h};
          layer_mac<8, 128, HidSrc<8>, BwdHook<8, 8, MASKS>, true>(acc, st, s, {h, dz.A(l), 256, act.A(l - 1), 256, mrow_of(kMaskGrpA + l - 2), raw, bits, sg});
        }
        mask_tiles<8>(acc, bits, h);
      }
      save_tiles<8>(h, dz.A(1), 256, sg);
      if constexpr (DX) {               // BL_X1: h = dZ1
        HidSrc<8> s{h};
        layer_mac<7, bwd_ks(BL_X1), HidSrc<8>, NoHook, false>(accx, st, s);
        if constexpr (!DO_OBJ) {
#pragma unroll
          for (int t = 0; t < 7; ++t) save_tile_cols<7>(accx, t, a.d_emb, a.ld_emb, kScnVoxPE, sg);
        }
      }
    }

    // ---------------- object branch ----------------
    if constexpr (DO_OBJ) {
      // keep the compiler from computing this branch's per-lane addresses ahead of the scene branch (spills)
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const Stage sg{sg_scene.buf, sg_scene.p0, sg_scene.P, lane_o};
      f32x16 acc[4], h[4];
      MaskBits<4> bits;
      {
        f32x16 hd[2];
        MaskBits<2> bd;
        if constexpr (MASKS) load_masks<2>(bd, mrow_of(kMaskGrpOD));
        else fetch_tiles<2>(raw, act.odirh(), 64, 0, sg);
        zero_tiles<2>(hd);
#pragma unroll
        for (int c = 0; c < 3; ++c) add_head<2>(hd, aux + kAuxORgb + c * 2 * 32, half, a.t2i[p * 3 + c]);
        if constexpr (!MASKS) sign_bits<2, 2, 0>(raw, bd, sg);
        mask_tiles<2>(hd, bd, hd);
        {
          HidSrc<2> s{hd};
          layer_mac<4, bwd_ks(BL_OD), HidSrc<2>, BwdHook<2, 4, MASKS>, true>(acc, st, s, {hd, dz.odirh(), 64, nullptr, 0, nullptr, raw, bits, sg});
        }
      }
      finish<4, false>(acc, h);
      {
        HidSrc<4> s{h};
        layer_mac<4, 64, HidSrc<4>, BwdHook<4, 4, MASKS>, true>(acc, st, s, {h, dz.ofinal(), 128, act.B(4), 128, mrow_of(kMaskGrpB + 3), raw, bits, sg});
      }
      add_head<4>(acc, aux + kAuxOSig, half, a.d_isigma[p]);
      mask_tiles<4>(acc, bits, h);
      f32x16 accxo[DX ? 4 : 1];         // d(object voxel embedding): 104 columns (4 tiles)
#pragma unroll 1
      for (int l = 4; l >= 2; --l) {
        if constexpr (DX) {
          if (l == 3) {                 // (uniform) BL_XS3, BL_XO3: h = dB3 meets the skip layer's scene- / object-voxel columns
            HidSrc<4> s{h};
            layer_mac<7, bwd_ks(BL_XS3), HidSrc<4>, NoHook, false>(accx, st, s);
            layer_mac<4, bwd_ks(BL_XO3), HidSrc<4>, NoHook, true>(accxo, st, s);
          }
        }
        {
          HidSrc<4> s{h};
          layer_mac<4, 64, HidSrc<4>, BwdHook<4, 4, MASKS>, true>(acc, st, s, {h, dz.B(l), 128, act.B(l - 1), 128, mrow_of(kMaskGrpB + l - 2), raw, bits, sg});
        }
        mask_tiles<4>(acc, bits, h);
      }
      save_tiles<4>(h, dz.B(1), 128, sg);
      if constexpr (DX) {               // BL_XS1, BL_XO1: h = dB1; then the finished embedding gradients leave the registers
        HidSrc<4> s{h};
        layer_mac<7, bwd_ks(BL_XS1), HidSrc<4>, NoHook, false>(accx, st, s);
        layer_mac<4, bwd_ks(BL_XO1), HidSrc<4>, NoHook, false>(accxo, st, s);
#pragma unroll
        for (int t = 0; t < 7; ++t) save_tile_cols<7>(accx, t, a.d_emb, a.ld_emb, kScnVoxPE, sg);
#pragma unroll
        for (int t = 0; t < 4; ++t) save_tile_cols<4>(accxo, t, a.d_ov, kObjVoxPE, kObjVoxPE, sg);
      }
    }
  }
}

long train_mask_floats_host(long n_points) { return train_mask_floats(n_points); }

int launch_mlp_bwd(const float* blob_bwd, const float* aux, long P, const float* act, float* dz, const float* d_sigma,
                   const float* t2, const float* d_isigma, const float* t2i, bool do_object, const unsigned* masks, bool dx,
                   float* d_emb, long ld_emb, float* d_ov, hipStream_t s) {
  static_assert(bwd_scene_chunks(false) == 68 || kChunkTiles != 128, "backward stream layout");
  // dx: the stream carries the embedding-gradient blocks (objnerf_pack_index_bwd mode 2) and the kernel walks them, i.e. forms
  // those gradients itself: needs the mask-fed chain and the two destinations
  if (dx && (!masks || !d_emb || (do_object && !d_ov)))
    return set_error(-1, "mlp_train_backward(fused): a blob_bwd with the embedding-gradient blocks needs the forward's masks, d_emb_xyz (and d_obj_voxel)");
  const BwdArgs a{blob_bwd, aux, P, act, dz, d_sigma, t2, d_isigma, t2i, masks, d_emb, d_ov, ld_emb};
  const long ntiles = (P + 127) / 128;
  const unsigned grid = mlp_grid(ntiles);
  if (dx) {
    if (do_object) hipLaunchKernelGGL((mlp_bwd_kernel<true, true, true>), dim3(grid), dim3(256), 0, s, a, ntiles);
    else hipLaunchKernelGGL((mlp_bwd_kernel<false, true, true>), dim3(grid), dim3(256), 0, s, a, ntiles);
  } else if (masks) {
    if (do_object) hipLaunchKernelGGL((mlp_bwd_kernel<true, true, false>), dim3(grid), dim3(256), 0, s, a, ntiles);
    else hipLaunchKernelGGL((mlp_bwd_kernel<false, true, false>), dim3(grid), dim3(256), 0, s, a, ntiles);
  } else {
    if (do_object) hipLaunchKernelGGL((mlp_bwd_kernel<true, false, false>), dim3(grid), dim3(256), 0, s, a, ntiles);
    else hipLaunchKernelGGL((mlp_bwd_kernel<false, false, false>), dim3(grid), dim3(256), 0, s, a, ntiles);
  }
  return check_launch("mlp_train_backward(fused)");
}

}  // namespace objnerf
