// layout.h -- K-order, weight-stream and aux layouts shared by the host-side packer
// and the gfx950 kernels.  Everything here is constexpr and host+device.
//
// Model dimensions are those of every shipped reference config
// (/root/reference/config/default_conf.yml:7-36): scene branch D=8 W=256 skips=[4],
// object branch inst_D=4 inst_W=128 inst_skips=[2], N_freq_xyz=10, N_freq_dir=4,
// N_freq_voxel=6, 16 scene + 8 object voxel channels, 64-d object code.
// (models/nerf_model.py:18-95 defines the layer shapes restated below.)
//
// The MLP is evaluated as  H_out^T[out, point] = W[out, k] * H_in^T[k, point]  with
// v_mfma_f32_32x32x2_f32:   A = W tile (32 out rows x 2 k), B = activations (2 k x 32
// points), D = 32 out rows x 32 points.  One wave owns 32 points.  Lane l holds point
// (l & 31); the two lane halves (l >> 5) supply the two k of one MFMA step.
//
// D layout (guide cdna_hip_programming.md §3):  col = lane & 31,
//   row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5),  reg in [0,16).
// Hence accumulator register `reg` of out-tile `t`, after bias+activation, is directly the
// B operand of k-step  ks = 16 t + reg  of the next layer, and that k-step contracts over
// the feature pair { hid_feat(ks, 0), hid_feat(ks, 1) }.  The packer permutes W's columns
// accordingly; activations never leave registers between layers.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define OBJ_HD __host__ __device__ inline
#else
#define OBJ_HD inline
#endif

namespace objnerf {

constexpr int kW = 256;          // scene width
constexpr int kIW = 128;         // object width
constexpr int kDirC = 27;        // 3 + 3*2*4
constexpr int kCodeC = 64;
constexpr int kScnVoxC = 16;
constexpr int kObjVoxC = 8;
constexpr int kVoxC = 24;
constexpr int kFreqXyz = 10;
constexpr int kFreqDir = 4;
constexpr int kFreqVox = 6;
constexpr int kXyzPE = 63;       // 3 + 3*2*10
constexpr int kScnVoxPE = kScnVoxC * (2 * kFreqVox + 1);   // 208
constexpr int kObjVoxPE = kObjVoxC * (2 * kFreqVox + 1);   // 104

// in_channels_xyz / inst_channel_in  (nerf_model.py:25-35, 62-72)
OBJ_HD constexpr int in_xyz(bool voxel) { return voxel ? kScnVoxPE + kXyzPE : kXyzPE; }          // 271 / 63
OBJ_HD constexpr int in_obj(bool voxel) { return in_xyz(voxel) + (voxel ? kObjVoxPE : 0) + kCodeC; }   // 439 / 127

// ---- per-lane-half K lists (number of k-steps; one k-step = 2 features, one per half) ----
constexpr int kKsXyz = 32;                       // 30 sin/cos + 2 raw slots
constexpr int kKsScnVox = 8 * 13;                // 8 channels per half * (1 + 12)
constexpr int kKsObjVox = 4 * 13;
constexpr int kKsCode = 32;
constexpr int kKsDir = 14;                       // 12 sin/cos + 2 raw slots
OBJ_HD constexpr int ks_emb(bool voxel) { return voxel ? kKsScnVox + kKsXyz : kKsXyz; }                 // 136 / 32
OBJ_HD constexpr int ks_objin(bool voxel) { return ks_emb(voxel) + (voxel ? kKsObjVox : 0) + kKsCode; } // 220 / 64

// feature index inside a hidden vector for k-step ks, lane half h
OBJ_HD constexpr int hid_feat(int ks, int h) {
  return 32 * (ks >> 4) + ((ks & 3) + 8 * ((ks & 15) >> 2)) + 4 * h;
}

// xyz positional-encoding slot i in [0,32) of half h -> column of Embedding(3,10) output
// (embedding_helper.py:69-74: [x, sin f0 x, cos f0 x, sin f1 x, ...], x = all 3 channels), -1 = pad
OBJ_HD constexpr int xyz_slot_col(int i, int h) {
  if (i < 30) {
    int p = i >> 1, fn = i & 1, coord = p / 5, k = 5 * h + p % 5;
    return 3 + (2 * k + fn) * 3 + coord;
  }
  if (i == 30) return h ? 2 : 0;
  return h ? -1 : 1;
}
// what to compute for xyz slot i: coord, frequency exponent (or -1 raw), fn 0 sin / 1 cos
OBJ_HD constexpr int xyz_slot_coord(int i, int h) { return i < 30 ? (i >> 1) / 5 : (i == 30 ? (h ? 2 : 0) : 1); }
OBJ_HD constexpr int xyz_slot_freq(int i, int h) { return i < 30 ? 5 * h + (i >> 1) % 5 : -1; }

// direction slot i in [0,14) of half h -> column of Embedding(3,4) output, -1 = pad
OBJ_HD constexpr int dir_slot_col(int i, int h) {
  if (i < 12) {
    int p = i >> 1, fn = i & 1, coord = p % 3, k = 2 * h + p / 3;
    return 3 + (2 * k + fn) * 3 + coord;
  }
  if (i == 12) return h ? 2 : 0;
  return h ? -1 : 1;
}

// voxel slot i in [0, nch*13) with nch channels per half out of C total channels:
// column inside Embedding(.,6) applied to a C-channel tensor (embedding_helper.py:409)
OBJ_HD constexpr int vox_slot_col(int i, int h, int nch, int C) {
  int fi = i / 13, j = i % 13, c = h * nch + fi;
  return j == 0 ? c : C + (j - 1) * C + c;
}

// scene embedding list: column of emb_xyz (EmbeddingVoxel.forward: cat([PE6(scene16), PE10(xyz)]),
// embedding_helper.py:325-329) for slot i of half h, -1 = pad
OBJ_HD constexpr int emb_slot_col(bool voxel, int i, int h) {
  if (!voxel) return xyz_slot_col(i, h);
  if (i < kKsScnVox) return vox_slot_col(i, h, 8, kScnVoxC);
  int c = xyz_slot_col(i - kKsScnVox, h);
  return c < 0 ? -1 : kScnVoxPE + c;
}
// object-branch input list: column of cat([emb_xyz, obj_voxel, obj_code]) (nerf_model.py:128-132)
OBJ_HD constexpr int objin_slot_col(bool voxel, int i, int h) {
  int ne = ks_emb(voxel);
  if (i < ne) return emb_slot_col(voxel, i, h);
  i -= ne;
  if (voxel) {
    if (i < kKsObjVox) return in_xyz(true) + vox_slot_col(i, h, 4, kObjVoxC);
    i -= kKsObjVox;
  }
  return in_xyz(voxel) + (voxel ? kObjVoxPE : 0) + h * 32 + i;
}

// ---- weight stream ------------------------------------------------------------------------
// One "A tile" = 64 floats (lane l: W[32 m + (l & 31)][kcol(ks, l >> 5)]).  A chunk = 128 A tiles
// = 32 KiB = (128 / NT) k-steps x NT out tiles, laid out [ks/4][m][lane][ks%4] so that one
// ds_read_b128 per out tile feeds 4 consecutive k-steps.  Every layer is padded to whole chunks.
#ifndef OBJ_CHUNK_TILES
#define OBJ_CHUNK_TILES 128
#endif
constexpr int kChunkTiles = OBJ_CHUNK_TILES;        // 128 -> 32 KiB chunks
constexpr int kChunkFloats = kChunkTiles * 64;
constexpr int kChunkBytes = kChunkFloats * 4;

enum LayerId {
  L_S1 = 0, L_S2, L_S3, L_S4, L_S5, L_S6, L_S7, L_S8, L_SF, L_SD,
  L_O1, L_O2, L_O3, L_O4, L_OF, L_OD, L_COUNT
};
OBJ_HD constexpr int layer_nt(int l) { return l <= L_SF ? 8 : (l == L_OD ? 2 : 4); }
OBJ_HD constexpr int layer_ks(bool voxel, int l) {
  switch (l) {
    case L_S1: return ks_emb(voxel);
    case L_S5: return ks_emb(voxel) + 128;
    case L_SD: return 128 + kKsDir;
    case L_O1: return ks_objin(voxel);
    case L_O3: return ks_objin(voxel) + 64;
    case L_OD: return 64 + kKsDir;
    default: return l <= L_SF ? 128 : 64;
  }
}
OBJ_HD constexpr int layer_chunks(bool voxel, int l) {
  int kg = kChunkTiles / layer_nt(l);
  return (layer_ks(voxel, l) + kg - 1) / kg;
}
OBJ_HD constexpr int layer_chunk_start(bool voxel, int l) {
  int s = 0;
  for (int i = 0; i < l; ++i) s += layer_chunks(voxel, i);
  return s;
}
OBJ_HD constexpr int scene_chunks(bool voxel) { return layer_chunk_start(voxel, L_O1); }
OBJ_HD constexpr int total_chunks(bool voxel) { return layer_chunk_start(voxel, L_COUNT); }

// column of the reference weight matrix for k-step ks, half h of layer l (-1 = zero pad)
OBJ_HD constexpr int layer_kcol(bool voxel, int l, int ks, int h) {
  const int ne = ks_emb(voxel), no = ks_objin(voxel);
  switch (l) {
    case L_S1: return emb_slot_col(voxel, ks, h);
    case L_S5:   // cat([input_xyz, h]) nerf_model.py:104-105
      return ks < ne ? emb_slot_col(voxel, ks, h) : in_xyz(voxel) + hid_feat(ks - ne, h);
    case L_SD: { // cat([xyz_encoding_final, input_dir]) nerf_model.py:116
      if (ks < 128) return hid_feat(ks, h);
      int c = dir_slot_col(ks - 128, h);
      return c < 0 ? -1 : kW + c;
    }
    case L_O1: return objin_slot_col(voxel, ks, h);
    case L_O3:   // cat([input_x, x_]) nerf_model.py:137-138
      return ks < no ? objin_slot_col(voxel, ks, h) : in_obj(voxel) + hid_feat(ks - no, h);
    case L_OD: { // cat([x_final, input_dir]) nerf_model.py:147
      if (ks < 64) return hid_feat(ks, h);
      int c = dir_slot_col(ks - 64, h);
      return c < 0 ? -1 : kIW + c;
    }
    default: return hid_feat(ks, h);
  }
}
OBJ_HD constexpr int layer_out(int l) { return l <= L_SF ? kW : (l == L_SD ? kIW : (l == L_OD ? kIW / 2 : kIW)); }
OBJ_HD constexpr int layer_in(bool voxel, int l) {
  switch (l) {
    case L_S1: return in_xyz(voxel);
    case L_S5: return in_xyz(voxel) + kW;
    case L_SD: return kW + kDirC;
    case L_O1: return in_obj(voxel);
    case L_O3: return in_obj(voxel) + kIW;
    case L_OD: return kIW + kDirC;
    default: return l <= L_SF ? kW : kIW;
  }
}

// ---- parameter ids (order of the device pointer table handed to the packer) ------------------
// scene: xyz_encoding_{1..8}.0, xyz_encoding_final, dir_encoding.0, sigma, rgb.0
// object: instance_encoding_{1..4}.0, instance_encoding_final.0, inst_dir_encoding.0,
//         instance_sigma, inst_rgb.0          (names: nerf_model.py:41-58, 77-95)
// weight pointer id = 2*p, bias pointer id = 2*p + 1
enum ParamId {
  P_S1 = 0, P_S2, P_S3, P_S4, P_S5, P_S6, P_S7, P_S8, P_SF, P_SD, P_SSIG, P_SRGB,
  P_O1, P_O2, P_O3, P_O4, P_OF, P_OD, P_OSIG, P_ORGB, P_COUNT
};
constexpr int kNumParamPtrs = 2 * P_COUNT;   // 40

// ---- aux block (biases + the tiny sigma / rgb heads, evaluated on the VALU) ---------------------
// bias of layer l: [m][half][16 regs] floats (value = bias[32 m + (r&3) + 8 (r>>2) + 4 half])
// sigma head (scene): [t(8)][half][16] weights then 1 bias;  rgb head: [c(3)][t(4)][half][16] then 3 biases
OBJ_HD constexpr int aux_bias_off(int l) {
  int s = 0;
  for (int i = 0; i < l; ++i) s += layer_nt(i) * 32;
  return s;
}
constexpr int kAuxSSig = 8 * 9 * 32 + 4 * 32 + 4 * 5 * 32 + 2 * 32;   // after all biases = 3136
constexpr int kAuxSRgb = kAuxSSig + 8 * 32 + 4;                        // 8 tiles * 32 + bias (padded to 4)
constexpr int kAuxOSig = kAuxSRgb + 3 * 4 * 32 + 4;
constexpr int kAuxORgb = kAuxOSig + 4 * 32 + 4;
constexpr int kAuxFloats = kAuxORgb + 3 * 2 * 32 + 4;
static_assert(aux_bias_off(L_COUNT) == kAuxSSig, "aux layout");

constexpr uint32_t kPackZero = 0xFFFFFFFFu;   // index-map entry meaning "0.0f"

// per-ray vectors of the hoisted terms (mlp_kernel HOIST, ray_bias_kernel): [O1 128 | O3 128 | SD 128 | OD 64] in the
// aux-bias layout of each layer
constexpr int kRayBiasFloats = 448;
// the weight columns those vectors are made from, as the A-operand stream of ray_bias_kernel (round 6: the per-ray vectors are
// a 448 x 91 product per ray -- 340 v_mfma_f32_32x32x2_f32 per 32 rays, exactly the MFMAs the hoisting takes out of the MLP
// kernel -- formed on the matrix pipe instead of 29 k fp32 FMAs per ray on the VALU): 14 out tiles T of 32 rows
//     T 0-3: instance_encoding_1 | 4-7: instance_encoding_3 (k = 64 code columns: 32 k-steps, column of (ks, h) = 32 h + ks)
//     T 8-11: dir_encoding | 12-13: inst_dir_encoding      (k = 27 direction columns: 14 k-steps, column of (ks, h) = 14 h + ks)
// each as [k-step group q of 4][lane][4 k-steps] floats (one ds_read_b128 feeds 4 MFMAs; lane = (row & 31) + 32 h), then the
// 448 biases in the per-ray vector's own order.  Gathered by objnerf_pack_models, kept behind the aux block
// (objnerf_aux_floats() = kAuxFloats + kRbMatFloats).
constexpr int kRbGroups = kRayBiasFloats / 16;                  // 28 groups of 16 floats: O1 8 | O3 8 | SD 8 | OD 4
constexpr int kRbCodeTiles = 8, kRbDirTiles = 6;
constexpr int kRbCodeQ = 8, kRbDirQ = 4;                        // k-step groups per tile (32 and 14 -> 16 k-steps, the last 2 zero)
constexpr int kRbDirKs = 14;
constexpr int kRbAFloats = (kRbCodeTiles * kRbCodeQ + kRbDirTiles * kRbDirQ) * 256;       // 22,528
constexpr int kRbMatFloats = kRbAFloats + kRayBiasFloats;
// first float of tile T in the stream, and its place (offset in the per-ray vector, tile index m inside its layer)
OBJ_HD constexpr int rb_tile_start(int T) { return (T < kRbCodeTiles ? T * kRbCodeQ : kRbCodeTiles * kRbCodeQ + (T - kRbCodeTiles) * kRbDirQ) * 256; }
OBJ_HD constexpr int rb_tile_off(int T) { return T < 4 ? 32 * T : (T < 8 ? 128 + 32 * (T - 4) : (T < 12 ? 256 + 32 * (T - 8) : 384 + 32 * (T - 12))); }

// ---- backward weight stream (training: dgrad of the hidden chain, mlp_bwd_kernel.h) ------------------
// d(input of layer) = W^T * d(pre-activation output): the same A-tile/chunk format with the roles swapped --
// tile row = INPUT feature 32 m + (lane & 31) of the streamed column block, k-step (ks, half) = OUTPUT feature
// hid_feat(ks, half), i.e. the order in which a wave holds the incoming gradient in its D-layout registers.
// Only hidden-to-hidden blocks are streamed (the gradients w.r.t. the embeddings are GEMMs, train.hip); in
// execution order:
enum BwdLayerId {
  BL_SD = 0, BL_SF, BL_S8, BL_S7, BL_S6, BL_X5, BL_S5, BL_S4, BL_S3, BL_S2, BL_X1,     // scene: dir hidden -> ... -> layer 1 output
  BL_OD, BL_OF, BL_O4, BL_XS3, BL_XO3, BL_O3, BL_O2, BL_XS1, BL_XO1, BL_COUNT      // object
};
// The X entries (round 6, voxel mode only: `dx` below) are the gradients w.r.t. the voxel-feature columns of the EMBEDDINGS, folded
// into the chain: while the gradient of a layer that consumes the embedding is in registers (dZ5, dZ1, dB3, dB1), the same B operand
// is contracted with that layer's embedding-column block -- X5 / X1: the 208 scene-voxel columns of xyz_encoding_5 / _1; XS3 / XS1:
// the same columns of instance_encoding_3 / _1; XO3 / XO1: their 104 object-voxel columns -- into accumulators that stay in
// registers until the end of the tile (7 + 4 tiles; rounds 2-5: two segmented GEMMs that re-read the four gradient matrices).
OBJ_HD constexpr bool bwd_is_x(int l) { return l == BL_X5 || l == BL_X1 || l == BL_XS3 || l == BL_XO3 || l == BL_XS1 || l == BL_XO1; }
OBJ_HD constexpr int bwd_nt(int l) {               // tiles of the produced gradient
  if (l == BL_XO3 || l == BL_XO1) return 4;
  if (bwd_is_x(l)) return 7;
  return l < BL_OD ? 8 : 4;
}
OBJ_HD constexpr int bwd_ks(int l) {
  if (l == BL_X5 || l == BL_X1) return kW / 2;
  if (bwd_is_x(l)) return kIW / 2;
  return l == BL_SD ? kIW / 2 : (l < BL_OD ? kW / 2 : (l == BL_OD ? kIW / 4 : kIW / 2));
}
// k-steps per chunk for a layer of nt out tiles: whole 4-k-step groups (7 tiles: 16 k-steps = 112 of the chunk's 128 tile slots)
OBJ_HD constexpr int chunk_ksteps(int nt) { return (kChunkTiles / nt) & ~3; }
OBJ_HD constexpr int bwd_chunks(int l) {
  int kg = chunk_ksteps(bwd_nt(l));
  return (bwd_ks(l) + kg - 1) / kg;
}
OBJ_HD constexpr int bwd_chunk_start(bool dx, int l) {
  int s = 0;
  for (int i = 0; i < l; ++i) s += (bwd_is_x(i) && !dx) ? 0 : bwd_chunks(i);
  return s;
}
OBJ_HD constexpr int bwd_scene_chunks(bool dx) { return bwd_chunk_start(dx, BL_OD); }
OBJ_HD constexpr int bwd_total_chunks(bool dx) { return bwd_chunk_start(dx, BL_COUNT); }
OBJ_HD constexpr int bwd_param(int l) {
  switch (l) {
    case BL_SD: return P_SD; case BL_SF: return P_SF;
    case BL_S8: return P_S8; case BL_S7: return P_S7; case BL_S6: return P_S6; case BL_S5: return P_S5;
    case BL_S4: return P_S4; case BL_S3: return P_S3; case BL_S2: return P_S2;
    case BL_X5: return P_S5; case BL_X1: return P_S1;
    case BL_OD: return P_OD; case BL_OF: return P_OF; case BL_O4: return P_O4; case BL_O3: return P_O3;
    case BL_XS3: case BL_XO3: return P_O3;
    case BL_XS1: case BL_XO1: return P_O1;
    default: return P_O2;
  }
}
// first column of the streamed block inside the reference weight matrix (the hidden block of a skip layer; the scene-voxel
// columns lead the embedding, the object-voxel columns follow its in_xyz columns, nerf_model.py:128-131) and the number of its rows
// that exist (the rest of the last tile is zero)
OBJ_HD constexpr int bwd_col0(bool voxel, int l) {
  if (l == BL_XO3 || l == BL_XO1) return in_xyz(voxel);
  if (bwd_is_x(l)) return 0;
  return l == BL_S5 ? in_xyz(voxel) : (l == BL_O3 ? in_obj(voxel) : 0);
}
OBJ_HD constexpr int bwd_rows(int l) {
  if (l == BL_XO3 || l == BL_XO1) return kObjVoxPE;
  if (bwd_is_x(l)) return kScnVoxPE;
  return 32 * bwd_nt(l);
}

}  // namespace objnerf
