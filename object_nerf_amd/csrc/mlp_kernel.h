// mlp_kernel.h -- fused encode + dual-branch MLP for gfx950 (MI355X).
//
// Replaces the reference's MLP chunk loop (models/rendering.py:106-130: EmbeddingVoxel.forward ->
// ObjectNeRF.forward -> ObjectNeRF.forward_instance, 20 GEMMs + ~700 elementwise launches per
// chunk) with ONE persistent kernel:
//
//   * workgroup = 4 waves (one per SIMD, up to 512 VGPR+AGPR each), 1 workgroup per CU;
//     each wave owns 32 consecutive sample points, the workgroup 128;
//   * activations never leave registers: layer l's 32x32 accumulator tiles (after bias +
//     LeakyReLU) ARE the B operands of layer l+1 (layout.h explains the permutation);
//   * weights are a linear stream of pre-permuted A tiles; 32 KiB chunks are DMA'd
//     global->LDS (buffer_load_dwordx4 ... lds) one chunk ahead into a 2-slot ring shared by the
//     4 waves -- issued piecewise between the MFMA groups of the chunk being consumed, not as one
//     burst -- and read back as ds_read_b128 (4 k-steps of one out tile per instruction);
//   * every XCD works on one contiguous eighth of the tiles (L2 locality of the voxel gathers);
//   * positional / voxel embeddings are generated in registers right where the MFMA needs
//     them (never materialised: the reference writes 375 floats per sample to HBM);
//   * the 1-row sigma heads and 3-row rgb heads run on the VALU (an MFMA tile would be
//     31/32 empty), combined across the two lane halves with one DPP/permute.
//
// Roofline: MFMA-bound. 13,876 v_mfma_f32_32x32x2_f32 per 32 points (both branches, voxel
// mode) = 1,776,128 algorithmic FLOP per point; fp32 MFMA peak 157.3 TFLOP/s.
//
// Template variants of the same kernel: FUSED = false reads pre-embedded rows (ObjectNeRF.forward /
// forward_instance), SIGMA_ONLY stops after the density head, SAVE is the training forward (every layer's
// output also written to memory).
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include <type_traits>
#include "layout.h"
#include "device_math.h"
#include "composite_seg.h"
#include "../../include/objnerf_hip.h"

// ---- tuning switches (defaults = the shipped configuration; tools/tune_mlp.py A/Bs them) ----
#ifndef OBJ_AUX_LDS
#define OBJ_AUX_LDS 1        // biases / head weights staged once per workgroup in LDS
#endif
#ifndef OBJ_OCTAVE_DOUBLING
#define OBJ_OCTAVE_DOUBLING 3   // P > 1: of every P octaves of one argument the first is evaluated in full, the P - 1 after it come
                                // from the double-angle identities; 0 or 1: every octave in full
#endif
#ifndef OBJ_PREFETCH_TILE
#define OBJ_PREFETCH_TILE 1  // gather prologue of the NEXT tile staged between the object-branch layers
#endif
#ifndef OBJ_PREFETCH_SCENE
#define OBJ_PREFETCH_SCENE 1 // density query, scene branch: gather prologue of the NEXT tile staged on the chunk barriers of xyz_encoding_1
#endif
#ifndef OBJ_SPREAD_DMA
#define OBJ_SPREAD_DMA 1     // weight DMA pieces issued between the MFMA groups instead of as a burst
#endif
#ifndef OBJ_NT_OUT
#define OBJ_NT_OUT 1         // sigma / rgb output stores carry the non-temporal hint
#endif
#ifndef OBJ_NT_ACT
#define OBJ_NT_ACT 1         // training kernels: activation stores / fetches carry the non-temporal hint
#endif
#ifndef OBJ_CODE_REGS
#define OBJ_CODE_REGS 1      // object code (32 floats per lane half) loaded once per pass into VGPRs
#endif

namespace objnerf {

typedef __attribute__((address_space(3))) char lds_char;

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// ---------------------------------------------------------------------------------------------
// weight stream: 2-slot LDS ring, one chunk (32 KiB) prefetched ahead, one barrier per chunk
// ---------------------------------------------------------------------------------------------
constexpr int kRingSlots = 2;   // the DMA runs one chunk ahead; vmcnt(0) is part of the chunk barrier

template <int CB>
struct WeightStreamT {
  static constexpr int kBytes = CB;   // bytes per chunk
  const char* win;     // global base of the stream window (first chunk of this mode)
  int nchunks;         // chunks in the window (wraps around: every pass replays it)
  int next;            // window index of the next chunk to DMA
  int cur;             // ring slot holding the chunk being consumed
  lds_char* ring;      // 2 * CB
  lds_char* rd;        // per-lane read base of the current slot (ring + cur*chunk + lane*16)
  int tid;
  __amdgpu_buffer_rsrc_t rsrc;
  int wave;            // wave-uniform (readfirstlane)

  // One chunk = CB, copied linearly global -> LDS by the 4 waves: wave w owns the contiguous quarter [w*Q, (w+1)*Q) and
  // moves it as Q/1024 buffer_load_dwordx4...lds: descriptor and chunk offset live in SGPRs, the piece offset in the
  // 12-bit immediate (applied to the global AND the LDS address), M0 (LDS base) changes once per
  // 4 KiB.  (A flat global_load_lds needs a 64-bit VALU address add + M0 write + hazard nop per
  // piece and measured ~49 cycles of lost MFMA issue per piece, 4 % of the kernel: docs/HISTORY_r1_r3.md.)
  template <int I>
  __device__ __forceinline__ void piece(lds_char* dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + (I >> 2) * 4096), 16,
                                             voff, soff + (I >> 2) * 4096, (I & 3) * 1024, 0);
  }
  __device__ __forceinline__ void issue(int slot) {
    // `next` is statically predictable inside one pass; hide it from the optimiser or LICM hoists
    // one 64-bit source address per (chunk, piece) out of the tile loop (hundreds of VGPRs, spills)
    int n = next;
    asm volatile("" : "+s"(n));
    constexpr int Q = CB / 4;
    lds_char* dst = ring + slot * CB + wave * Q;
    const int soff = n * CB + wave * Q;
    const int voff = (tid & 63) * 16;
    static_for<Q / 1024>([&](auto I) __attribute__((always_inline)) { piece<decltype(I)::value>(dst, voff, soff); });
    next = (next + 1 == nchunks) ? 0 : next + 1;
  }
  // Spread mode: next_chunk() only selects the chunk; its kPieces DMA instructions are then issued one at a time
  // between the MFMA groups of the chunk being consumed (layer_mac).  Issued as one burst, the 4 waves' 32 KiB take
  // ~500 cycles to drain through the 64 B/clk vector-memory path and stall the issuing waves for that long.
  static constexpr int kPieces = CB / 4 / 1024;       // per wave and chunk
  lds_char* pend_dst;
  int pend_soff;
  __device__ __forceinline__ void select(int slot) {
    int n = next;
    asm volatile("" : "+s"(n));
    pend_dst = ring + slot * CB + wave * (CB / 4);
    pend_soff = n * CB + wave * (CB / 4);
    next = (next + 1 == nchunks) ? 0 : next + 1;
  }
  template <int I>
  __device__ __forceinline__ void piece_now() {
    if constexpr (I < kPieces) piece<I>(pend_dst, (tid & 63) * 16, pend_soff);
  }
  __device__ __forceinline__ void init(const char* w, int n, lds_char* r, int t) {
    win = w; nchunks = n; next = 0; ring = r; tid = t; cur = kRingSlots - 1;
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, n * CB, 0x00020000);
    wave = __builtin_amdgcn_readfirstlane(t >> 6);
    issue(0);
  }
  // called right before the first A read of a chunk
  __device__ __forceinline__ void next_chunk() {
    __syncthreads();   // all DMA of the chunk landed (vmcnt(0) is part of the barrier) and every
                       // wave is done reading the slot we are about to overwrite
    cur ^= 1;
    if constexpr (OBJ_SPREAD_DMA) select(cur ^ 1);
    else issue(cur ^ 1);
    rd = ring + cur * CB + (tid & 63) * 16;
  }
};

using WeightStream = WeightStreamT<kChunkBytes>;

__device__ __forceinline__ f32x4 lds_read16(const lds_char* p) {
  return *(const __attribute__((address_space(3))) f32x4*)p;
}

// acc[m] += W_tile(m, ks) * B(ks) for every k-step of one layer.  Src::get<ks>() yields this
// lane's B operand (feature (ks, lane>>5) of point lane&31).
template <int NT>
struct ATiles { f32x4 v[NT]; };

template <int NT, int G, class Stream>
__device__ __forceinline__ void load_group(ATiles<NT>& a, Stream& st) {
  constexpr int KG = chunk_ksteps(NT);      // (= kChunkTiles / NT for 1, 2, 4, 8 tiles)
  constexpr int ks0 = G * 4;
  if constexpr (ks0 % KG == 0) st.next_chunk();
  constexpr int g4 = (ks0 % KG) / 4;
#pragma unroll
  for (int m = 0; m < NT; ++m) a.v[m] = lds_read16(st.rd + (g4 * NT + m) * 1024);
}

// Software pipeline, one group (4 k-steps x NT out tiles) per stage: the A tiles of group g+1 are
// read from LDS (after the chunk barrier when g+1 opens a new chunk) before the MFMAs of group g
// issue.  sched_barrier(0) between stages keeps the compiler from hoisting the embedding
// arithmetic of later groups (it would otherwise keep hundreds of sin/cos values live and spill).
// `after_barrier(integral_constant<int, c>)` runs right behind the barrier that opens the layer's c-th chunk.  The
// barrier drains vmcnt (LDS-DMA), so global loads/stores issued just BEFORE it are waited for in full, while ones
// issued just AFTER it have a whole chunk of MFMAs to complete: the training kernels put their activation stores and
// prefetches there.
struct NoHook {
  static constexpr int min_groups = 0;      // groups the layer must have for the hook to have stored everything
  template <int C> __device__ __forceinline__ void operator()(std::integral_constant<int, C>) const {}
  template <int GI> __device__ __forceinline__ void group() const {}     // in front of the layer's GI-th MFMA group
};
// ZERO: acc = W * B instead of acc += W * B (the first k-step's MFMA takes the constant 0 as its C operand; the
// accumulators need no initialisation pass).
template <int A, int B> struct SkipGroups { static constexpr int k0 = A, k1 = B; };
using NoSkip = SkipGroups<0, 0>;
// SK0, SK1: the 4-k-step groups [SK0, SK1) of the layer are SKIPPED -- no A reads, no MFMAs -- while the weight stream
// keeps its schedule (their chunks are opened and the next chunk's DMA pieces issued as usual).  Used when the terms of
// those k-steps are constant along a ray and arrive through objnerf_mlp_args.ray_bias instead (HOIST, see mlp_kernel).
template <int NT, int KS, class Src, class Hook = NoHook, bool ZERO = false, class Stream = WeightStream, class SkipT = NoSkip>
__device__ __forceinline__ void layer_mac(f32x16 (&acc)[NT], Stream& st, Src& src, Hook after_barrier = Hook{}, SkipT = SkipT{}) {
  constexpr int SK0 = SkipT::k0, SK1 = SkipT::k1;
  static_assert(SK0 == SK1 || SK0 > 0, "group 0 is never skipped");
  constexpr int NG4 = (KS + 3) / 4;
  static_assert(NG4 >= Hook::min_groups, "the layer is shorter than its hook's store schedule");
  constexpr int KG = chunk_ksteps(NT);      // (= kChunkTiles / NT for 1, 2, 4, 8 tiles)
  // spread mode: the 8 DMA pieces of the chunk after the current one are issued in front of the chunk's MFMA groups
  constexpr int GPC = KG / 4;                                  // groups per chunk
  // pieces per group, front-loaded into the first half of the chunk's groups: the chunk is opened (barrier, vmcnt(0))
  // in front of the current chunk's LAST group, so the last piece gets at least a third of a chunk to land
  constexpr int PPG = (Stream::kPieces + (GPC / 2 > 0 ? GPC / 2 : 1) - 1) / (GPC / 2 > 0 ? GPC / 2 : 1);
  auto pieces = [&](auto GQ) __attribute__((always_inline)) {
    if constexpr (OBJ_SPREAD_DMA)
      static_for<PPG>([&](auto Q) __attribute__((always_inline)) { st.template piece_now<decltype(GQ)::value * PPG + decltype(Q)::value>(); });
  };
  ATiles<NT> abuf[2];
  load_group<NT, 0>(abuf[0], st);
  after_barrier(std::integral_constant<int, 0>{});
  static_for<NG4>([&](auto G) __attribute__((always_inline)) {
    constexpr int g = decltype(G)::value;
    constexpr int ks0 = g * 4;
    ATiles<NT>& a = abuf[g & 1];
    pieces(std::integral_constant<int, g % GPC>{});              // group g's share (after the barrier that opened its chunk)
    after_barrier.template group<g>();
    constexpr bool skipped = g >= SK0 && g < SK1;
    if constexpr (g + 1 < NG4) {
      constexpr bool next_skipped = g + 1 >= SK0 && g + 1 < SK1;
      if constexpr (!next_skipped) load_group<NT, g + 1>(abuf[(g + 1) & 1], st);
      else if constexpr (((g + 1) * 4) % KG == 0) st.next_chunk();        // a skipped group still opens its chunk
      if constexpr (((g + 1) * 4) % KG == 0) after_barrier(std::integral_constant<int, ((g + 1) * 4) / KG>{});
    } else if constexpr (OBJ_SPREAD_DMA) {
      // end of the layer: pieces the (shorter) last chunk had no group for
      static_for<Stream::kPieces>([&](auto I) __attribute__((always_inline)) {
        if constexpr (decltype(I)::value >= (g % GPC + 1) * PPG) st.template piece_now<decltype(I)::value>();
      });
    }
    static_for<4>([&](auto J) __attribute__((always_inline)) {
      constexpr int j = decltype(J)::value;
      constexpr int ks = ks0 + j;
      if constexpr (ks < KS && !skipped) {
        const float b = src.template get<ks>();
#pragma unroll
        for (int m = 0; m < NT; ++m) {
          if constexpr (ZERO && ks == 0) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[m][j], b, zero, 0, 0, 0);
          } else {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[m][j], b, acc[m], 0, 0, 0);
          }
        }
      }
    });
    __builtin_amdgcn_sched_barrier(0);
  });
}

template <int NT>
__device__ __forceinline__ void load_bias(f32x16 (&acc)[NT], const float* aux, int layer, int half) {
  const float* b = aux + aux_bias_off(layer) + half * 16;
#pragma unroll
  for (int m = 0; m < NT; ++m) acc[m] = *(const f32x16*)(b + m * 32);
}

#ifndef OBJ_PK_LEAKY
#define OBJ_PK_LEAKY 1
#endif
using f32x2 = __attribute__((ext_vector_type(2))) float;
template <int NT, bool ACT>
__device__ __forceinline__ void finish(const f32x16 (&acc)[NT], f32x16 (&h)[NT]) {
#pragma unroll
  for (int m = 0; m < NT; ++m)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      if constexpr (ACT && OBJ_PK_LEAKY) {
        // 0.01 v for two values in one v_pk_mul_f32 (gfx950 has packed fp32 mul / fma, no packed max): the same two IEEE
        // operations per value as leaky() -- bit-equal -- in 1.5 instead of 2 VALU instructions
        const f32x2 v = {acc[m][r], acc[m][r + 1]};
        const f32x2 sl = v * 0.01f;
        h[m][r] = fmaxf(v[0], sl[0]);
        h[m][r + 1] = fmaxf(v[1], sl[1]);
      } else {
        h[m][r] = ACT ? leaky(acc[m][r]) : acc[m][r];
        h[m][r + 1] = ACT ? leaky(acc[m][r + 1]) : acc[m][r + 1];
      }
    }
}

// HOIST: per-ray vectors (objnerf_mlp_args.ray_bias, aux-bias layout [m][half][16]) and the epilogue that adds them
constexpr int kRbO1 = 0, kRbO3 = 128, kRbSD = 256, kRbOD = 384;
#ifndef OBJ_NT_TABLE
#define OBJ_NT_TABLE 1       // feature-table rows with the non-temporal hint as well (A/B: profiles/r04_nt_ab.txt)
#endif
#ifndef OBJ_NT_IDX
#define OBJ_NT_IDX 0         // ... and the corner -> row index map
#endif
__device__ __forceinline__ f32x4 table_load(const float* p) {
#if OBJ_NT_TABLE
  return __builtin_nontemporal_load((const f32x4*)p);
#else
  return *(const f32x4*)p;
#endif
}
template <class T>
__device__ __forceinline__ T idx_load(const T* p) {
#if OBJ_NT_IDX
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
#ifndef OBJ_NT_RAYVEC
#define OBJ_NT_RAYVEC 1      // per-ray vectors (550 MB per pass, read once per wave) fetched with the non-temporal hint
#endif
template <int NT>
__device__ __forceinline__ void load_rb(f32x16 (&dst)[NT], const float* p) {
#pragma unroll
  for (int m = 0; m < NT; ++m) {
#if OBJ_NT_RAYVEC
    // streamed once per wave: must not push the weight stream (3.55 MB of the XCD's 4 MB L2, replayed by every tile) out --
    // a 0.4 % miss rate of THAT stream is what the memory-side counters mostly see (profiles/r04_pmc.md)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = __builtin_nontemporal_load((const f32x4*)(p + m * 32 + 4 * q));
      dst[m][4 * q] = v[0]; dst[m][4 * q + 1] = v[1]; dst[m][4 * q + 2] = v[2]; dst[m][4 * q + 3] = v[3];
    }
#else
    dst[m] = *(const f32x16*)(p + m * 32);
#endif
  }
}
template <int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NT]) {
#pragma unroll
  for (int m = 0; m < NT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
}
// h = act(acc + add); `add` may alias h
template <int NT, bool ACT>
__device__ __forceinline__ void finish_add(const f32x16 (&acc)[NT], const f32x16 (&add)[NT], f32x16 (&h)[NT]) {
#pragma unroll
  for (int m = 0; m < NT; ++m)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      if constexpr (OBJ_PK_LEAKY) {      // packed add (+ packed 0.01 v): the same IEEE operations, half the instructions
        const f32x2 a2 = {acc[m][r], acc[m][r + 1]}, b2 = {add[m][r], add[m][r + 1]};
        const f32x2 v = a2 + b2;
        if constexpr (ACT) {
          const f32x2 sl = v * 0.01f;
          h[m][r] = fmaxf(v[0], sl[0]);
          h[m][r + 1] = fmaxf(v[1], sl[1]);
        } else {
          h[m][r] = v[0];
          h[m][r + 1] = v[1];
        }
      } else {
        const float v0 = acc[m][r] + add[m][r], v1 = acc[m][r + 1] + add[m][r + 1];
        h[m][r] = ACT ? leaky(v0) : v0;
        h[m][r + 1] = ACT ? leaky(v1) : v1;
      }
    }
}

// dot of this lane's NT*16 hidden features with a packed head row, summed over both halves
template <int NT>
__device__ __forceinline__ float head_dot(const f32x16 (&h)[NT], const float* w, int half) {
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < NT; ++m) {
    const f32x16 wv = *(const f32x16*)(w + (m * 2 + half) * 16);
#pragma unroll
    for (int r = 0; r < 16; ++r) s = fmaf(h[m][r], wv[r], s);
  }
  return s + __shfl_xor(s, 32);
}

// ---------------------------------------------------------------------------------------------
// B-operand sources
// ---------------------------------------------------------------------------------------------
template <int NT>
struct HidSrc {
  const f32x16 (&h)[NT];
  template <int I>
  __device__ __forceinline__ float get() { return h[I >> 4][I & 15]; }
};

// Fused source: embeds the point in registers.  VOXEL: voxel-grid mode (EmbeddingVoxel), else
// plain Embedding(3,10).
template <bool VOXEL>
struct FusedSrc {
  float vf[12];        // trilinear voxel features of this half: 8 scene ch (half*8+i), 4 object ch
  float pos[3];        // sample position
  float dir[3];        // ray direction
  float fscale;        // 2^(5*half): xyz frequency split
  float dscale;        // 2^(2*half): dir frequency split
  const float* code;   // this ray's object code + half*32
#if OBJ_CODE_REGS
  f32x4 codev[8];      // the 32 code values of this half, fetched at the top of the object branch
#endif
  int half;
  float saved_cos, saved_sin;

  __device__ __forceinline__ void fetch_code() {
#if OBJ_CODE_REGS
#pragma unroll
    for (int i = 0; i < 8; ++i) codev[i] = *(const f32x4*)(code + 4 * i);
#endif
  }

  // Makes the embedding inputs opaque to the optimiser.  Called before every layer that consumes
  // the embedding: without it LLVM's GVN reuses the sin/cos values of the first consumer for the
  // later ones (S1 -> S5 -> O1 -> O3), keeps 136..220 values live across whole layers and spills.
  // Recomputing costs ~1 % of the MFMA time; spilling costs scratch traffic inside the MFMA loops.
  __device__ __forceinline__ void launder() {
#pragma unroll
    for (int i = 0; i < 12; ++i) asm volatile("" : "+v"(vf[i]));
#pragma unroll
    for (int i = 0; i < 3; ++i) { asm volatile("" : "+v"(pos[i])); asm volatile("" : "+v"(dir[i])); }
  }
  // positional-encoding pair slots: even local index = sin, odd = cos of the same argument;
  // slots are consumed in increasing order, so the cos rides along from the sin slot
  __device__ __forceinline__ float pe_pair(float arg, int fn) {
    if (fn == 0) { const SinCos sc = psincos(arg); saved_cos = sc.c; saved_sin = sc.s; return sc.s; }
    return saved_cos;
  }
  // Octave k of the SAME base argument, consumed in increasing k (k = 0 restarts).  Every third octave
  // is evaluated in full; the two after it come from the double-angle identities
  //   sin 2a = 2 sin a cos a,  cos 2a = 1 - 2 sin^2 a        (4 VALU instead of ~26)
  // whose absolute error at most doubles per step: <= ~6e-7 after two steps, f32-roundoff class.
  // fp32 MFMA and VALU share the SIMD's ALUs, so this arithmetic is paid in MFMA time (4.8 % of the
  // kernel before this change).
  template <int K>
  __device__ __forceinline__ float pe_octave(float base, int fn) {
#if OBJ_OCTAVE_DOUBLING > 1
    if constexpr (K % OBJ_OCTAVE_DOUBLING != 0) {
      if (fn == 0) {
        const float s = saved_sin, c = saved_cos;
        saved_sin = (2.f * s) * c;
        saved_cos = fmaf(-2.f * s, s, 1.f);
        return saved_sin;
      }
      return saved_cos;
    }
#endif
    return pe_pair(base * (float)(1 << K), fn);
  }
  template <int I>
  __device__ __forceinline__ float xyz_slot() {
    if constexpr (I < 30) {
      constexpr int p = I >> 1, coord = p / 5, kk = p % 5;
      return pe_octave<kk>(pos[coord] * fscale, I & 1);
    } else if constexpr (I == 30) {
      return half ? pos[2] : pos[0];
    } else {
      return half ? 0.f : pos[1];
    }
  }
  template <int I, int BASE>
  __device__ __forceinline__ float vox_slot() {
    constexpr int fi = I / 13, j = I % 13;
    if constexpr (j == 0) return vf[BASE + fi];
    else {
      constexpr int k = (j - 1) >> 1;
      return pe_octave<k>(vf[BASE + fi], (j - 1) & 1);
    }
  }
  template <int I>
  __device__ __forceinline__ float emb() {
    if constexpr (VOXEL) {
      if constexpr (I < kKsScnVox) return vox_slot<I, 0>();
      else return xyz_slot<I - kKsScnVox>();
    } else {
      return xyz_slot<I>();
    }
  }
  template <int I>
  __device__ __forceinline__ float objin() {
    constexpr int ne = ks_emb(VOXEL);
    if constexpr (I < ne) return emb<I>();
    else if constexpr (VOXEL && I < ne + kKsObjVox) return vox_slot<I - ne, 8>();
    else {
      constexpr int ci = I - ne - (VOXEL ? kKsObjVox : 0);
#if OBJ_CODE_REGS
      return codev[ci >> 2][ci & 3];
#else
      return code[ci];
#endif
    }
  }
  template <int I>
  __device__ __forceinline__ float dirslot() {
    if constexpr (I < 12) {
      constexpr int p = I >> 1, coord = p % 3, kk = p / 3;
      return pe_pair(dir[coord] * (float)(1 << kk) * dscale, I & 1);
    } else if constexpr (I == 12) {
      return half ? dir[2] : dir[0];
    } else {
      return half ? 0.f : dir[1];
    }
  }
};

// Memory source: pre-embedded rows exactly as ObjectNeRF.forward / forward_instance get them
template <bool VOXEL>
struct MemSrc {
  const float* exyz;   // row of emb_xyz  (in_xyz)
  const float* edir;   // row of emb_dir  (27)
  const float* ovox;   // row of obj_voxel (104) or null
  const float* ocode;  // row of obj_code (64)
  int half;
  // same purpose as FusedSrc::launder: keep GVN from carrying the S1/O1 loads to S5/O3 (spills)
  __device__ __forceinline__ void launder() {
    asm volatile("" : "+v"(exyz));
    asm volatile("" : "+v"(edir));
    asm volatile("" : "+v"(ovox));
    asm volatile("" : "+v"(ocode));
    // `half ? c1 : c0` column selects are invariant across the persistent tile loop: without this
    // LICM hoists one offset VGPR per K slot out of the loop (hundreds of registers, spills)
    asm volatile("" : "+v"(half));
  }
  __device__ __forceinline__ void fetch_code() {}
  __device__ __forceinline__ float pick(const float* row, int c0, int c1) {
    const int c = half ? c1 : c0;
    return c < 0 ? 0.f : row[c < 0 ? 0 : c];
  }
  template <int I>
  __device__ __forceinline__ float emb() {
    return pick(exyz, emb_slot_col(VOXEL, I, 0), emb_slot_col(VOXEL, I, 1));
  }
  template <int I>
  __device__ __forceinline__ float objin() {
    constexpr int ne = ks_emb(VOXEL);
    if constexpr (I < ne) return emb<I>();
    else if constexpr (VOXEL && I < ne + kKsObjVox) {
      constexpr int ii = I - ne;
      return pick(ovox, vox_slot_col(ii, 0, 4, kObjVoxC), vox_slot_col(ii, 1, 4, kObjVoxC));
    } else {
      constexpr int ii = I - ne - (VOXEL ? kKsObjVox : 0);
      return ocode[half * 32 + ii];
    }
  }
  template <int I>
  __device__ __forceinline__ float dirslot() {
    return pick(edir, dir_slot_col(I, 0), dir_slot_col(I, 1));
  }
};

// adaptors: one layer's concatenated K list
template <class S> struct EmbOnly { S& s; template <int I> __device__ __forceinline__ float get() { return s.template emb<I>(); } };
template <class S> struct ObjInOnly { S& s; template <int I> __device__ __forceinline__ float get() { return s.template objin<I>(); } };
template <class S, int NE, int NT> struct EmbThenHid {
  S& s; const f32x16 (&h)[NT];
  template <int I> __device__ __forceinline__ float get() {
    if constexpr (I < NE) return s.template emb<I>(); else return h[(I - NE) >> 4][(I - NE) & 15];
  }
};
template <class S, int NE, int NT> struct ObjInThenHid {
  S& s; const f32x16 (&h)[NT];
  template <int I> __device__ __forceinline__ float get() {
    if constexpr (I < NE) return s.template objin<I>(); else return h[(I - NE) >> 4][(I - NE) & 15];
  }
};
template <class S, int NH, int NT> struct HidThenDir {
  S& s; const f32x16 (&h)[NT];
  template <int I> __device__ __forceinline__ float get() {
    if constexpr (I < NH) return h[I >> 4][I & 15]; else return s.template dirslot<I - NH>();
  }
};

// ---------------------------------------------------------------------------------------------
// voxel feature fetch: EmbeddingVoxel.compute_voxel_features_sparse (embedding_helper.py:354-411)
// restricted to the 12 channels of this lane half.  Operation order mirrors the reference:
// s = (xyz + offset) / voxel_size (IEEE divide), q = floor(s), p = s - q, corner weights as the
// products (a*b)*c, sum over the 8 corners in itertools.product([0,1],repeat=3) order.
//
// per-tile gather prologue, in stages.  Run back to back it is the plain prologue; with OBJ_PREFETCH_TILE the
// stages of the NEXT tile are placed between the object-branch layers of the current one, so that each level
// of its dependent loads (ray row + depth -> 8 index-map reads -> 24 feature-row reads) has a whole layer of
// MFMAs to land (the chain cost 1.5 % of the kernel when it ran at the top of every pass).
// Arithmetic and its order are identical in both placements.
// ---------------------------------------------------------------------------------------------
// POINTS (the fused sigma-only form, objnerf_mlp_args.points / lat_*): the tile's 128 points are given directly -- an
// explicit (n, 3) array or the nodes of a lattice in np.meshgrid(x, y, z) order -- instead of rays x depths; there is no
// direction and no per-ray quantity.
template <bool VOXEL, bool POINTS = false>
struct TilePrologue {
  long p, ray;
  int sidx;            // the point's sample index inside its ray
  bool valid;
  float rw[8];         // ray row [o, d, near, far]
  float zv;
  float pos[3], dir[3];
  float w[8];          // trilinear corner weights
  int idx[8];          // index-map entries (or -1: out of the grid)
  f32x4 rows[12];      // feature rows in flight: 4 corners x {scene lo, scene hi, object} quarter rows of this half
  float vf[12];

  __device__ __forceinline__ void stage_a(const objnerf_mlp_args& a, long tile, long P, int wave, int lane) {
    const long p_raw = tile * 128 + wave * 32 + (lane & 31);
    valid = p_raw < P;
    const long pc = valid ? p_raw : P - 1;
    if constexpr (POINTS) {
      p = pc; ray = 0; sidx = 0; zv = 0.f;
      if (a.points) {
        const float* q = a.points + pc * 3;
        rw[0] = q[0]; rw[1] = q[1]; rw[2] = q[2];
      } else {
        // np.stack(np.meshgrid(x, y, z), -1).reshape(-1, 3) (tools/extract_mesh.py:62-66, indexing 'xy'):
        // point ((j * nx + i) * nz + k) = (x[i], y[j], z[k])
        const long t = pc / a.lat_n[2];
        const int k = (int)(pc - t * a.lat_n[2]);
        const long j = t / a.lat_n[0];
        const int i = (int)(t - j * a.lat_n[0]);
        rw[0] = a.lat_x[i]; rw[1] = a.lat_y[j]; rw[2] = a.lat_z[k];
      }
      rw[3] = rw[4] = rw[5] = rw[6] = rw[7] = 0.f;
      return;
    }
    const long slot = pc / a.S;
    sidx = (int)(pc - slot * a.S);                                   // the point's sample index inside its ray
    // ray subset (objnerf_mlp_args.ray_index): tiles walk the listed rays only; p stays the point's index in the
    // full (n_rays, S) arrays, so depths are read and results written in place
    ray = a.ray_index ? (long)a.ray_index[slot] : slot;
    p = ray * a.S + sidx;
    const float* r = a.rays + ray * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) rw[i] = r[i];
    zv = a.z_vals[p];
  }
  __device__ __forceinline__ void stage_b(const objnerf_voxel_grid& g) {
    // xyz = rays_o + rays_d * z  (rendering.py:279): separately rounded mul and add
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dir[c] = rw[3 + c];
      pos[c] = POINTS ? rw[c] : rw[c] + dir[c] * zv;
    }
    if constexpr (VOXEL) {
      const float sx = __fdiv_rn(pos[0] + g.offset[0], g.voxel_size);
      const float sy = __fdiv_rn(pos[1] + g.offset[1], g.voxel_size);
      const float sz = __fdiv_rn(pos[2] + g.offset[2], g.voxel_size);
      const float qx = floorf(sx), qy = floorf(sy), qz = floorf(sz);
      const float u = sx - qx, v = sy - qy, ww = sz - qz;
      const float lu = 1.f - u, lv = 1.f - v, lw = 1.f - ww;
      w[0] = (lu * lv) * lw; w[1] = (lu * lv) * ww; w[2] = (lu * v) * lw; w[3] = (lu * v) * ww;
      w[4] = (u * lv) * lw;  w[5] = (u * lv) * ww;  w[6] = (u * v) * lw;  w[7] = (u * v) * ww;
      const float X = (float)g.shape[0], Y = (float)g.shape[1], Z = (float)g.shape[2];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float cx = qx + (float)((k >> 2) & 1), cy = qy + (float)((k >> 1) & 1), cz = qz + (float)(k & 1);
        const bool ok = cx >= 0.f && cx < X && cy >= 0.f && cy < Y && cz >= 0.f && cz < Z;
        idx[k] = ok ? idx_load(g.idx_map + (((size_t)(int)cx * g.shape[1] + (int)cy) * g.shape[2] + (int)cz)) : -1;
      }
    }
  }
  // issue the feature-row reads of corners [4*H, 4*H+4)
  template <int H>
  __device__ __forceinline__ void stage_rows(const objnerf_voxel_grid& g, int half) {
    if constexpr (VOXEL) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = idx[4 * H + j];
        const int row = (r < 0 || r >= g.n_rows) ? -1 : r;
        idx[4 * H + j] = row;
        const float* t = g.table + (size_t)(row < 0 ? 0 : row) * kVoxC;
        rows[3 * j + 0] = table_load(t + half * 8);
        rows[3 * j + 1] = table_load(t + half * 8 + 4);
        rows[3 * j + 2] = table_load(t + kScnVoxC + half * 4);
      }
    }
  }
  // voxel_ftr[invalid] = 0 ; (voxel_ftr * weights).sum(0)   (embedding_helper.py:351,387-389), corners in order
  template <int H>
  __device__ __forceinline__ void stage_acc() {
    if constexpr (VOXEL) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 4 * H + j;
        const bool bad = idx[k] < 0;
        const float wk = w[k];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float f0 = bad ? 0.f : rows[3 * j][i], f1 = bad ? 0.f : rows[3 * j + 1][i], f2 = bad ? 0.f : rows[3 * j + 2][i];
          if (k == 0) { vf[i] = f0 * wk; vf[4 + i] = f1 * wk; vf[8 + i] = f2 * wk; }
          else { vf[i] = vf[i] + f0 * wk; vf[4 + i] = vf[4 + i] + f1 * wk; vf[8 + i] = vf[8 + i] + f2 * wk; }
        }
      }
    }
  }
  // the same two corners at a time (rows[0..5]: 24 registers in flight instead of 48) for the scene-only variants, whose
  // next-tile prologue rides on the chunk barriers of xyz_encoding_1 (PrefetchHook)
  template <int Q>
  __device__ __forceinline__ void stage_rows2(const objnerf_voxel_grid& g, int half) {
    if constexpr (VOXEL) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = idx[2 * Q + j];
        const int row = (r < 0 || r >= g.n_rows) ? -1 : r;
        idx[2 * Q + j] = row;
        const float* t = g.table + (size_t)(row < 0 ? 0 : row) * kVoxC;
        rows[3 * j + 0] = table_load(t + half * 8);
        rows[3 * j + 1] = table_load(t + half * 8 + 4);
        rows[3 * j + 2] = table_load(t + kScnVoxC + half * 4);
      }
    }
  }
  template <int Q>
  __device__ __forceinline__ void stage_acc2() {
    if constexpr (VOXEL) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int k = 2 * Q + j;
        const bool bad = idx[k] < 0;
        const float wk = w[k];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float f0 = bad ? 0.f : rows[3 * j][i], f1 = bad ? 0.f : rows[3 * j + 1][i], f2 = bad ? 0.f : rows[3 * j + 2][i];
          if (k == 0) { vf[i] = f0 * wk; vf[4 + i] = f1 * wk; vf[8 + i] = f2 * wk; }
          else { vf[i] = vf[i] + f0 * wk; vf[4 + i] = vf[4 + i] + f1 * wk; vf[8 + i] = vf[8 + i] + f2 * wk; }
        }
      }
    }
  }
  __device__ __forceinline__ void run_all(const objnerf_mlp_args& a, long tile, long P, int wave, int lane, int half) {
    stage_a(a, tile, P, wave, lane);
    stage_b(a.grid);
    stage_rows<0>(a.grid, half);
    stage_acc<0>();
    stage_rows<1>(a.grid, half);
    stage_acc<1>();
  }
};

// layer_mac hook of xyz_encoding_1 in the scene-only variants (no object branch to stage the next tile's prologue under):
// one level of the dependent chain per chunk barrier -- ray row / point, 8 index-map reads, then the feature rows two
// corners at a time -- each consumed behind the NEXT barrier (3.4 us of MFMAs later; the barrier's vmcnt(0) drains them
// anyway).  xyz_encoding_1 is the one layer during which no hidden vector is live: 128 registers to spare.
template <bool VOXEL, bool POINTS>
struct PrefetchHook {
  TilePrologue<VOXEL, POINTS>& pre;
  const objnerf_mlp_args& a;
  long tile, P;
  int wave, lane, half;
  template <int C>
  __device__ __forceinline__ void operator()(std::integral_constant<int, C>) const {
    if constexpr (C == 0) pre.stage_a(a, tile, P, wave, lane);
    if constexpr (C == 1) pre.stage_b(a.grid);
    if constexpr (VOXEL) {
      if constexpr (C == 2) pre.template stage_rows2<0>(a.grid, half);
      if constexpr (C == 3) { pre.template stage_acc2<0>(); pre.template stage_rows2<1>(a.grid, half); }
      if constexpr (C == 4) { pre.template stage_acc2<1>(); pre.template stage_rows2<2>(a.grid, half); }
      if constexpr (C == 5) { pre.template stage_acc2<2>(); pre.template stage_rows2<3>(a.grid, half); }
      if constexpr (C == 6) pre.template stage_acc2<3>();
    }
  }
  static constexpr int min_groups = 0;
  template <int GI> __device__ __forceinline__ void group() const {}
};

// ---------------------------------------------------------------------------------------------
// LeakyReLU masks of the training forward (round 5).  The dgrad chain (mlp_bwd.hip) needs only the SIGN of every saved hidden
// activation, in the D layout its gradient tiles have -- which is the layout the forward holds them in.  Rounds 3-4 re-read the
// row-major activation matrices (1.1 GB per launch), turned them around through the wave's LDS patch and compared; now the
// forward packs the signs of each lane's own registers (one v_cmp + one v_addc per value: the carry shifts the bit in) and
// stores 16 bytes per lane and layer, and the chain loads them back with one 16-byte load per layer.
// Word w[wd] of a layer: bit 16 (t & 1) + r = (h[t][r] > 0) for the tiles t = 2 wd, 2 wd + 1 -- the order mask_tiles consumes.
// Memory: [wave tile = point / 32][group][lane][4 dwords], groups A1..A8 | dir hidden | B1..B4 | object dir hidden.
// ---------------------------------------------------------------------------------------------
constexpr int kMaskGroups = 14;
constexpr int kMaskGrpA = 0, kMaskGrpSD = 8, kMaskGrpB = 9, kMaskGrpOD = 13;
constexpr long kMaskDwordsPerWave = (long)kMaskGroups * 64 * 4;           // 14 KiB per 32 points = 448 B per point
OBJ_HD constexpr long train_mask_floats(long n_points) { return ((n_points + 127) / 128) * 4 * kMaskDwordsPerWave; }
template <int NT>
struct MaskBits { unsigned w[NT / 2]; };     // tile t -> bits 16 (t & 1) .. + 15 of w[t / 2]
// word WD of the mask of h: tiles 2 WD + 1 (first: it ends in the upper half) and 2 WD, registers 15 .. 0
template <int NT, int WD>
__device__ __forceinline__ void sign_word(const f32x16 (&h)[NT], MaskBits<NT>& b) {
  unsigned m = 0;
#pragma unroll
  for (int tt = 1; tt >= 0; --tt)
#pragma unroll
    for (int r = 15; r >= 0; --r)
      asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(h[2 * WD + tt][r]) : "vcc");
  b.w[WD] = m;
}
template <int NT>
__device__ __forceinline__ unsigned* mask_row(unsigned* mask_ws, long p0, int grp, int lane) {
  return mask_ws + (p0 >> 5) * kMaskDwordsPerWave + ((long)grp * 64 + lane) * 4;
}
template <int NT>
__device__ __forceinline__ void store_masks(const MaskBits<NT>& b, unsigned* row) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  if constexpr (NT == 8) __builtin_nontemporal_store(u32x4{b.w[0], b.w[1], b.w[2], b.w[3]}, (u32x4*)row);
  else if constexpr (NT == 4) __builtin_nontemporal_store(u32x2{b.w[0], b.w[1]}, (u32x2*)row);
  else __builtin_nontemporal_store(b.w[0], row);
}
template <int NT>
__device__ __forceinline__ void load_masks(MaskBits<NT>& b, const unsigned* row) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  if constexpr (NT == 8) { const u32x4 v = __builtin_nontemporal_load((const u32x4*)row); b.w[0] = v[0]; b.w[1] = v[1]; b.w[2] = v[2]; b.w[3] = v[3]; }
  else if constexpr (NT == 4) { const u32x2 v = __builtin_nontemporal_load((const u32x2*)row); b.w[0] = v[0]; b.w[1] = v[1]; }
  else b.w[0] = __builtin_nontemporal_load(row);
}
// all words at once (the two direction layers, whose output is stored right behind their epilogue)
template <int NT>
__device__ __forceinline__ void sign_store_all(const f32x16 (&h)[NT], unsigned* row) {
  MaskBits<NT> b;
  static_for<NT / 2>([&](auto W) __attribute__((always_inline)) { sign_word<NT, decltype(W)::value>(h, b); });
  store_masks<NT>(b, row);
}

// ---------------------------------------------------------------------------------------------
// training forward (SAVE): every layer's output also goes to the row-major (P x width) activation matrices the
// layer-wise backward reads (train.hip, struct Ws).  h[t][4g..4g+3] of lane half hf are features
// 32 t + 8 g + 4 hf .. + 3 of one point.
// ---------------------------------------------------------------------------------------------
// The lane-native form of a tile (16 bytes per lane at a 1 KB stride between points) costs one memory request per
// 16-byte piece -- measured: a quarter of the training kernels' time.  Each wave therefore turns the tile around in a
// private 32 x 36-float LDS patch: D layout in (ds_write_b128, conflict-free at the 36-float row stride), point rows
// out (ds_read_b128), so that 8 consecutive lanes store one full 128-byte line of a point.
constexpr int kStageLd = 36;
constexpr int kStageFloats = 32 * kStageLd;         // one wave
constexpr int kStageBytes = 4 * kStageFloats * 4;   // four waves
struct Stage {
  float* buf;     // this wave's patch (LDS)
  long p0;        // first point of the wave's 32
  long P;
  int lane;
};
template <int NT>
__device__ __forceinline__ void save_tile(const f32x16 (&h)[NT], int t, float* mat, long ld, const Stage& sg) {
  const int pt = sg.lane & 31, half = sg.lane >> 5, q = sg.lane >> 3, k = sg.lane & 7;
  {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v = {h[t][4 * g], h[t][4 * g + 1], h[t][4 * g + 2], h[t][4 * g + 3]};
      *(f32x4*)(sg.buf + pt * kStageLd + 8 * g + 4 * half) = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 8 * i + q;
      const f32x4 v = *(const f32x4*)(sg.buf + row * kStageLd + 4 * k);
      // non-temporal: ~10 GB of activations stream past per training step and must not evict the 2-4 MB weight
      // stream every workgroup replays from L2
      if (sg.p0 + row < sg.P) {
#if OBJ_NT_ACT
        __builtin_nontemporal_store(v, (f32x4*)(mat + (sg.p0 + row) * ld + 32 * t + 4 * k));
#else
        *(f32x4*)(mat + (sg.p0 + row) * ld + 32 * t + 4 * k) = v;
#endif
      }
    }
  }
}
template <int NT>
__device__ __forceinline__ void save_tiles(const f32x16 (&h)[NT], float* mat, long ld, const Stage& sg) {
#pragma unroll
  for (int t = 0; t < NT; ++t) save_tile<NT>(h, t, mat, ld, sg);
}
template <bool ON, int NT>
struct SaveHook {      // layer_mac after-barrier hook: write h (the layer's input = the previous layer's output)
  const f32x16 (&h)[NT];
  float* mat;
  long ld;
  const Stage& sg;
  unsigned* mrow;      // this lane's 16 bytes of h's LeakyReLU mask (nullptr: h is not the output of an activated layer)
  MaskBits<NT>& mb;
  template <int C>
  __device__ __forceinline__ void operator()(std::integral_constant<int, C>) const {}
  // one tile per MFMA group (a burst of all NT tiles' stores stalls the issuing wave like a burst of DMA pieces does); the
  // mask word of a tile pair behind the pair's second tile, the mask store behind the last one.
  // Round 6 measured what the saves cost and three other ways to do them (profiles/r06_train_probe_save.txt): without staging and
  // stores the forward and the dgrad chain run 7 % faster, without the global stores alone 1.3 %; lane-native 16-byte pieces
  // straight from the D layout (no LDS patch), every 4th group (away from the chunk barriers), and patch write / patch read /
  // store spread over three groups all measured the same as this form -- none is kept.
  static constexpr int min_groups = ON ? NT : 0;
  template <int GI>
  __device__ __forceinline__ void group() const {
    if constexpr (ON && GI < NT) {
      save_tile<NT>(h, GI, mat, ld, sg);
      if (mrow) {
        if constexpr (GI & 1) sign_word<NT, GI / 2>(h, mb);
        if constexpr (GI == NT - 1) store_masks<NT>(mb, mrow);
      }
    }
  }
};
// saved-activation matrices, floats per point: scene 8 x 256 | final 256 | dir hidden 128 | (4 unused) |
// object 4 x 128 | final 128 | dir hidden 64 | (4 unused)   -- same arithmetic as train.hip's Ws
struct SaveWs {
  float* base;
  long P;
  __device__ __forceinline__ float* A(int l) const { return base + (long)(l - 1) * 256 * P; }     // l = 1..8
  __device__ __forceinline__ float* sfinal() const { return base + 8L * 256 * P; }
  __device__ __forceinline__ float* sdirh() const { return sfinal() + 256L * P; }
  __device__ __forceinline__ float* B(int l) const { return sdirh() + 128L * P + 4L * P + (long)(l - 1) * 128 * P; }   // l = 1..4
  __device__ __forceinline__ float* ofinal() const { return B(5); }
  __device__ __forceinline__ float* odirh() const { return ofinal() + 128L * P; }
};

// sigma / rgb results: written once, read by the compositing kernel after this one has finished -- non-temporal, so
// that 1.9 GB per frame of outputs do not push the weight stream and the voxel rows out of the XCD's L2
__device__ __forceinline__ void out_store(float* p, float v) {
#if OBJ_NT_OUT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
// HOIST: the terms of the direction layers (27 of 283 / 155 inputs) and of the object branch's input layers (the 64-d code:
// 64 of 439 / 567 inputs) that are CONSTANT ALONG A RAY -- the direction embedding and the object code are per-ray
// quantities the reference repeats over the samples (rendering.py:89-94) -- are not contracted per sample: their k-steps
// are skipped (340 of 13,876 MFMAs per 32 points, 2.45 %) and  bias + W[:, those columns] . x  arrives once per ray from
// ray_bias_kernel (objnerf_ray_bias), added in the layer's epilogue.  Same sums in another association: fp32-roundoff class.
template <bool VOXEL, bool FUSED, bool DO_SCENE, bool DO_OBJ, bool SIGMA_ONLY = false, bool SAVE = false, bool HOIST = false>
__global__ void __launch_bounds__(256, 1) mlp_kernel(const objnerf_mlp_args a, const long ntiles_arg, float* const save_ws = nullptr,
                                                     unsigned* const mask_ws = nullptr) {
  // (with SIGMA_ONLY: the object-branch density query, whose ONE code is constant over all points -- its share of
  // instance_encoding_1 / _3 arrives as a single vector at ray_bias, every point reads "ray" 0)
  // (SAVE + HOIST, round 5: the training forward hoists as well -- the saved activations and masks are those of the same layers)
  static_assert(!HOIST || (FUSED && (!SIGMA_ONLY || (DO_OBJ && !DO_SCENE))), "hoisting: the fused kernel");
  constexpr int kCB = kChunkBytes;       // bytes per weight chunk
  static_assert(!SIGMA_ONLY || (DO_SCENE != DO_OBJ), "sigma-only: one branch per launch (contiguous stream window)");
  static_assert(!SAVE || !SIGMA_ONLY, "the training forward needs every layer");
  // ONE __shared__ object (a second one makes hipcc drain vmcnt before every ds_read of a glds
  // pipeline, guide §5 "three .s-level traps"): [2-slot weight ring | aux block]
  __shared__ __attribute__((aligned(16))) char ring_mem[kRingSlots * kCB + (OBJ_AUX_LDS ? kAuxFloats * 4 : 0) +
                                                        (SAVE ? kStageBytes : 0)];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int wave = tid >> 6;

  constexpr int kStart = DO_SCENE ? 0 : scene_chunks(VOXEL);
  constexpr int kEnd = SIGMA_ONLY ? layer_chunk_start(VOXEL, DO_SCENE ? L_SF : L_OF)
                                  : (DO_OBJ ? total_chunks(VOXEL) : scene_chunks(VOXEL));
  // ray subset: the number of listed rays lives in device memory (written by objnerf_compact_rays earlier on the
  // stream), so a caller can cull rays without a host round trip; the grid was sized for all n_rays
  constexpr bool POINTS = FUSED && SIGMA_ONLY;       // the density query on explicit points / a lattice
  long P = (FUSED && !POINTS) ? a.n_rays * (long)a.S : a.n_points;
  long ntiles = ntiles_arg;
  if constexpr (FUSED && !SAVE) {
    if (a.n_active) {
      P = (long)(*a.n_active) * a.S;
      ntiles = (P + 127) / 128;
      if (P == 0) return;            // uniform across the grid; nothing has been issued yet
    }
  }
  // Tile -> workgroup map.  Workgroups are dealt round-robin to the 8 XCDs (workgroup b runs on XCD b % 8), each with
  // its own 4 MB L2.  Every XCD gets ONE contiguous eighth of the tiles and its 32 workgroups walk it side by side, so
  // the voxel-table rows shared by neighbouring depths / neighbouring pixels are fetched into one L2 instead of eight
  // (with the plain "tile = b + k * grid" map consecutive tiles land on 8 different XCDs).
  const bool by_xcd = (gridDim.x & 7) == 0;
  const long tiles_per_xcd = (ntiles + 7) / 8;
  const long tile_first = by_xcd ? (blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3) : blockIdx.x;
  const long tile_step = by_xcd ? (gridDim.x >> 3) : gridDim.x;
  const long tile_end = by_xcd ? (((blockIdx.x & 7) + 1) * tiles_per_xcd < ntiles ? ((blockIdx.x & 7) + 1) * tiles_per_xcd : ntiles)
                               : ntiles;
  // a workgroup without a tile (the grid is sized for all n_rays, a culled ray subset may need far fewer): leave before
  // the weight DMA, the aux staging and the first gather prologue are issued -- uniform per workgroup
  if (tile_first >= tile_end) return;

  WeightStreamT<kCB> st;
  st.init((const char*)a.blob + (size_t)kStart * kCB, kEnd - kStart,
          (lds_char*)ring_mem, tid);

  const SaveWs ws{save_ws, P};
#if OBJ_AUX_LDS
  // biases + head weights (16 KiB) are read by every wave every pass: stage them once in LDS
  float* aux_lds = (float*)(ring_mem + kRingSlots * kCB);
  for (int i = tid; i < kAuxFloats; i += 256) aux_lds[i] = a.aux[i];
  __syncthreads();
  const float* aux = aux_lds;
#else
  const float* aux = a.aux;
#endif

  using Src = std::conditional_t<FUSED, FusedSrc<VOXEL>, MemSrc<VOXEL>>;
  constexpr int NE = ks_emb(VOXEL);
  constexpr int NO = ks_objin(VOXEL);

  constexpr bool PREFETCH = OBJ_PREFETCH_TILE && FUSED && DO_OBJ;
  // scene-branch density query: there is no object branch to hide the next tile's gather chain under -- it rides on the
  // chunk barriers of xyz_encoding_1 (PrefetchHook).  Not for the other scene-only variants (BASELINE configs[0], the
  // background set of render_rays_multi): the next tile's 24 values would be live across the direction / colour layers
  // too, and those variants sit at the 256 architectural registers already (measured at compile time: 48 spills).
  static_assert(ks_emb(VOXEL) / (kChunkTiles / 8) + (ks_emb(VOXEL) % (kChunkTiles / 8) != 0) >= (VOXEL ? 7 : 2),
                "xyz_encoding_1 must have a chunk per prologue stage");
  constexpr bool PREFETCH_S = OBJ_PREFETCH_TILE && OBJ_PREFETCH_SCENE && POINTS && DO_SCENE && !DO_OBJ;
  TilePrologue<VOXEL, POINTS> pre;
  if constexpr (FUSED) {
    pre.run_all(a, tile_first, P, wave, lane, half);
  }
  // compositing in the epilogue (objnerf_mlp_args.comp_w, composite_seg.h): inference form of the fused kernel only
  constexpr bool COMP = FUSED && !SAVE && !SIGMA_ONLY && DO_SCENE;
  for (long tile = tile_first; tile < tile_end; tile += tile_step) {
    long p;
    bool valid;
    float comp_z = 0.f, comp_zn = 0.f, comp_sg = 0.f, comp_c[3] = {0.f, 0.f, 0.f};
    bool comp_last = false;
    const float* rbp = nullptr;      // HOIST: this lane's ray vectors (+ half * 16)
    Src src;
    src.half = half;
    if constexpr (FUSED) {
      if constexpr (!PREFETCH && !PREFETCH_S) {
        if (tile != tile_first) {
          pre.run_all(a, tile, P, wave, lane, half);
        }
      }
      p = pre.p;
      valid = pre.valid;
      if constexpr (HOIST) rbp = a.ray_bias + pre.ray * kRayBiasFloats + half * 16;
      if constexpr (COMP) {
        // compositing in the epilogue: this sample's depth and the next one's (the prologue registers are re-used for
        // the NEXT tile during the object branch); one extra load per tile, consumed a whole pass later
        if (a.comp_w) {
          comp_z = pre.zv;
          comp_last = pre.sidx + 1 >= a.S;
          comp_zn = comp_last ? 0.f : a.z_vals[p + 1];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) { src.dir[c] = pre.dir[c]; src.pos[c] = pre.pos[c]; }
#pragma unroll
      for (int i = 0; i < 12; ++i) src.vf[i] = pre.vf[i];
      src.fscale = half ? 32.f : 1.f;
      src.dscale = half ? 4.f : 1.f;
      src.code = DO_OBJ ? a.codes + pre.ray * a.code_stride + half * 32 : nullptr;
    } else {
      const long p_raw = tile * 128 + wave * 32 + (lane & 31);
      valid = p_raw < P;
      p = valid ? p_raw : P - 1;
      src.exyz = a.emb_xyz + p * in_xyz(VOXEL);
      src.edir = a.emb_dir + p * kDirC;
      src.ovox = (VOXEL && DO_OBJ) ? a.obj_voxel + p * kObjVoxPE : nullptr;
      src.ocode = DO_OBJ ? a.obj_code + p * kCodeC : nullptr;
    }
    const Stage stg{(float*)(ring_mem + kRingSlots * kCB + (OBJ_AUX_LDS ? kAuxFloats * 4 : 0)) + wave * kStageFloats,
                   tile * 128 + wave * 32, P, lane};
    // tile whose prologue is staged during this pass (the last pass re-stages its own tile: harmless)
    const long tile_next = tile + tile_step < tile_end ? tile + tile_step : tile;

    if constexpr (DO_SCENE) {
      f32x16 acc[8], h[8];
      // xyz_encoding_1
      src.launder();
      load_bias<8>(acc, aux, L_S1, half);
      if constexpr (PREFETCH_S) {
        EmbOnly<Src> s{src};
        layer_mac<8, NE>(acc, st, s, PrefetchHook<VOXEL, POINTS>{pre, a, tile_next, P, wave, lane, half});
      } else {
        EmbOnly<Src> s{src};
        layer_mac<8, NE>(acc, st, s);
      }
      finish<8, true>(acc, h);
      // SAVE: a layer's output is written by the NEXT layer's after-barrier hook (see layer_mac); h is that layer's
      // input and stays live anyway
      // (grp: the mask group of h, or -1 when h is not the output of an activated layer)
      MaskBits<8> mb;
      auto save_h = [&](float* mat, int grp) __attribute__((always_inline)) {
        return SaveHook<SAVE, 8>{h, mat, 256, stg, (SAVE && mask_ws && grp >= 0) ? mask_row<8>(mask_ws, stg.p0, grp, lane) : nullptr, mb};
      };
      // xyz_encoding_2..4
#pragma unroll 1
      for (int l = L_S2; l <= L_S4; ++l) {
        load_bias<8>(acc, aux, l, half);
        { HidSrc<8> s{h}; layer_mac<8, 128>(acc, st, s, save_h(ws.A(l - L_S1), kMaskGrpA + l - L_S1 - 1)); }
        finish<8, true>(acc, h);
      }
      // xyz_encoding_5 (skip: cat([emb, h]))
      src.launder();
      load_bias<8>(acc, aux, L_S5, half);
      { EmbThenHid<Src, NE, 8> s{src, h}; layer_mac<8, NE + 128>(acc, st, s, save_h(ws.A(4), kMaskGrpA + 3)); }
      finish<8, true>(acc, h);
#pragma unroll 1
      for (int l = L_S6; l <= L_S8; ++l) {
        load_bias<8>(acc, aux, l, half);
        { HidSrc<8> s{h}; layer_mac<8, 128>(acc, st, s, save_h(ws.A(l - L_S1), kMaskGrpA + l - L_S1 - 1)); }
        finish<8, true>(acc, h);
      }
      // sigma head (no activation, nerf_model.py:108)
      const float sg = head_dot<8>(h, aux + kAuxSSig, half) + aux[kAuxSSig + 8 * 32];
      if constexpr (SIGMA_ONLY) {
        if (valid && half == 0) out_store(a.sigma + p, sg);
      } else {
      // xyz_encoding_final (no activation)
      load_bias<8>(acc, aux, L_SF, half);
      { HidSrc<8> s{h}; layer_mac<8, 128>(acc, st, s, save_h(ws.A(8), kMaskGrpA + 7)); }
      finish<8, false>(acc, h);
      // dir_encoding: cat([final, dir]) -> W/2, LeakyReLU
      f32x16 acc4[4], hd[4];
      if constexpr (HOIST) {
        // bias + W[:, 256:283] . PE(dir) comes per ray; the 14 direction k-steps (groups 32..35) are skipped
        load_rb<4>(hd, rbp + kRbSD);
        zero_acc<4>(acc4);
        { HidSrc<8> s{h}; layer_mac<4, 128 + kKsDir>(acc4, st, s, save_h(ws.sfinal(), -1), SkipGroups<32, (128 + kKsDir + 3) / 4>{}); }
        finish_add<4, true>(acc4, hd, hd);
      } else {
        src.launder();
        load_bias<4>(acc4, aux, L_SD, half);
        { HidThenDir<Src, 128, 8> s{src, h}; layer_mac<4, 128 + kKsDir>(acc4, st, s, save_h(ws.sfinal(), -1)); }
        finish<4, true>(acc4, hd);
      }
      if constexpr (SAVE) {
        save_tiles<4>(hd, ws.sdirh(), 128, stg);
        if (mask_ws) sign_store_all<4>(hd, mask_row<4>(mask_ws, stg.p0, kMaskGrpSD, lane));
      }
      float col[3];
#pragma unroll
      for (int c = 0; c < 3; ++c)
        col[c] = sigmoidf(head_dot<4>(hd, aux + kAuxSRgb + c * 4 * 32, half) + aux[kAuxSRgb + 3 * 4 * 32 + c]);
      if (valid && half == 0 && a.sigma) {
        out_store(a.sigma + p, sg);
        if (a.rgb) { out_store(a.rgb + p * 3 + 0, col[0]); out_store(a.rgb + p * 3 + 1, col[1]); out_store(a.rgb + p * 3 + 2, col[2]); }
      }
      if constexpr (COMP) { comp_sg = sg; comp_c[0] = col[0]; comp_c[1] = col[1]; comp_c[2] = col[2]; }
      }
    }

    if constexpr (DO_OBJ) {
      f32x16 acc[4], h[4];
      // code k-steps of the object input list: the last kKsCode of its NO k-steps = whole groups [NOC / 4, NO / 4)
      constexpr int NOC = NO - kKsCode;
      static_assert(NOC % 4 == 0 && NO % 4 == 0, "the code block is a whole number of 4-k-step groups");
      using SkipCode = SkipGroups<NOC / 4, NO / 4>;
      if constexpr (!HOIST) src.fetch_code();      // 8 x 16-B loads, consumed ~190 k-steps later: latency fully hidden
      if constexpr (PREFETCH) pre.stage_a(a, tile_next, P, wave, lane);
      src.launder();
      if constexpr (HOIST) {
        load_rb<4>(h, rbp + kRbO1);                // bias + W[:, code columns] . code, per ray; consumed by the epilogue
        zero_acc<4>(acc);
        { ObjInOnly<Src> s{src}; layer_mac<4, NO>(acc, st, s, NoHook{}, SkipCode{}); }
        finish_add<4, true>(acc, h, h);
      } else {
        load_bias<4>(acc, aux, L_O1, half);
        { ObjInOnly<Src> s{src}; layer_mac<4, NO>(acc, st, s); }
        finish<4, true>(acc, h);
      }
      MaskBits<4> mb;
      auto save_h = [&](float* mat, int grp) __attribute__((always_inline)) {
        return SaveHook<SAVE, 4>{h, mat, 128, stg, (SAVE && mask_ws && grp >= 0) ? mask_row<4>(mask_ws, stg.p0, grp, lane) : nullptr, mb};
      };
      if constexpr (PREFETCH) pre.stage_b(a.grid);
      load_bias<4>(acc, aux, L_O2, half);
      { HidSrc<4> s{h}; layer_mac<4, 64>(acc, st, s, save_h(ws.B(1), kMaskGrpB)); }
      finish<4, true>(acc, h);
      if constexpr (PREFETCH) pre.template stage_rows<0>(a.grid, half);
      src.launder();
      if constexpr (HOIST) {
        f32x16 t3[4];                              // h is this layer's input: the ray vector waits in registers of its own
        load_rb<4>(t3, rbp + kRbO3);
        zero_acc<4>(acc);
        { ObjInThenHid<Src, NO, 4> s{src, h}; layer_mac<4, NO + 64>(acc, st, s, save_h(ws.B(2), kMaskGrpB + 1), SkipCode{}); }
        finish_add<4, true>(acc, t3, h);
      } else {
        load_bias<4>(acc, aux, L_O3, half);
        { ObjInThenHid<Src, NO, 4> s{src, h}; layer_mac<4, NO + 64>(acc, st, s, save_h(ws.B(2), kMaskGrpB + 1)); }
        finish<4, true>(acc, h);
      }
      if constexpr (PREFETCH) { pre.template stage_acc<0>(); pre.template stage_rows<1>(a.grid, half); }
      load_bias<4>(acc, aux, L_O4, half);
      { HidSrc<4> s{h}; layer_mac<4, 64>(acc, st, s, save_h(ws.B(3), kMaskGrpB + 2)); }
      finish<4, true>(acc, h);
      const float sg = head_dot<4>(h, aux + kAuxOSig, half) + aux[kAuxOSig + 4 * 32];
      if constexpr (PREFETCH) pre.template stage_acc<1>();
      if constexpr (SIGMA_ONLY) {
        if (valid && half == 0) out_store(a.inst_sigma + p, sg);
      } else {
      load_bias<4>(acc, aux, L_OF, half);
      { HidSrc<4> s{h}; layer_mac<4, 64>(acc, st, s, save_h(ws.B(4), kMaskGrpB + 3)); }
      finish<4, false>(acc, h);
      f32x16 acc2[2], hd[2];
      if constexpr (HOIST) {
        load_rb<2>(hd, rbp + kRbOD);
        zero_acc<2>(acc2);
        { HidSrc<4> s{h}; layer_mac<2, 64 + kKsDir>(acc2, st, s, save_h(ws.ofinal(), -1), SkipGroups<16, (64 + kKsDir + 3) / 4>{}); }
        finish_add<2, true>(acc2, hd, hd);
      } else {
        src.launder();
        load_bias<2>(acc2, aux, L_OD, half);
        { HidThenDir<Src, 64, 4> s{src, h}; layer_mac<2, 64 + kKsDir>(acc2, st, s, save_h(ws.ofinal(), -1)); }
        finish<2, true>(acc2, hd);
      }
      if constexpr (SAVE) {
        save_tiles<2>(hd, ws.odirh(), 64, stg);
        if (mask_ws) sign_store_all<2>(hd, mask_row<2>(mask_ws, stg.p0, kMaskGrpOD, lane));
      }
      float col[3];
#pragma unroll
      for (int c = 0; c < 3; ++c)
        col[c] = sigmoidf(head_dot<2>(hd, aux + kAuxORgb + c * 2 * 32, half) + aux[kAuxORgb + 3 * 2 * 32 + c]);
      if (valid && half == 0 && a.inst_sigma) {
        out_store(a.inst_sigma + p, sg);
        if (a.inst_rgb) { out_store(a.inst_rgb + p * 3 + 0, col[0]); out_store(a.inst_rgb + p * 3 + 1, col[1]); out_store(a.inst_rgb + p * 3 + 2, col[2]); }
      }
      if constexpr (COMP) {
        // instance set of this wave's 32-sample segment (last delta 0, rendering.py:148,213); both lane halves hold the
        // same 32 points, the lower half's totals are kept
        if (a.comp_w && valid) {
          const float delta = comp_last ? 0.f : comp_zn - comp_z;
          SegTotals lo, hi;
          const float lw = segment_composite(sample_alpha(delta, sg), true, col[0], col[1], col[2], comp_z, lane, lo, hi);
          if (a.comp_inst_weights && half == 0) a.comp_w[p] = lw;
          if (lane == 0) {
            float* rec = a.comp_rec + (p >> 5) * kSegRecFloats + kSegRecInst;
            *(f32x4*)rec = f32x4{lo.Q, lo.A, lo.R, lo.G};
            rec[4] = lo.B; rec[5] = lo.D;
          }
        }
      }
      }
    }
    if constexpr (COMP) {
      // scene set (last delta 1e10, or 0 with use_zero_as_last_delta: rendering.py:143-153)
      if (a.comp_w && valid) {
        const float delta = comp_last ? a.comp_last_delta : comp_zn - comp_z;
        SegTotals lo, hi;
        const float lw = segment_composite(sample_alpha(delta, comp_sg), true, comp_c[0], comp_c[1], comp_c[2], comp_z, lane, lo, hi);
        if (!(DO_OBJ && a.comp_inst_weights) && half == 0) a.comp_w[p] = lw;
        if (lane == 0) {
          float* rec = a.comp_rec + (p >> 5) * kSegRecFloats;
          *(f32x4*)rec = f32x4{lo.Q, lo.A, lo.R, lo.G};
          rec[4] = lo.B; rec[5] = lo.D;
        }
      }
    }
  }
}

}  // namespace objnerf
