// composite_seg.h -- the alpha-compositing arithmetic (models/rendering.py:139-229) in the form BOTH of its users share:
//
//   * composite_kernel (ray_kernels.hip, "K3"): one wave per ray, sigma / rgb read from memory -- training flags
//     (occlusion mask, noise), odd sample counts, the stage entry point objnerf_composite;
//   * the fused MLP kernel's epilogue (mlp_kernel.h, objnerf_mlp_args.comp_*) + composite_finish_kernel: sigma / rgb of
//     both branches never leave the registers they were computed in (eval mode, S a multiple of 32).
//
// A ray's samples are cut into SEGMENTS of 32 consecutive samples.  Per segment (32 lanes, DPP steps only):
//     t_i  = (1 - alpha_i) + 1e-10                      alphas_shifted, rendering.py:159-161
//     e_i  = prod_{k < i, same segment} t_k             exclusive scan
//     lw_i = alpha_i * e_i                              the weight the sample would have if the segment started the ray
//     Q = prod t_i;  A = sum lw_i;  C = sum lw_i c_i;  D = sum lw_i z_i
// and per ray, segments in ascending order with T_0 = 1:
//     w_i = T_j * lw_i;  opacity += T_j * A_j;  rgb += T_j * C_j;  depth += T_j * D_j;  T_{j+1} = T_j * Q_j.
// Same products and sums as the reference's cumprod / sum, associated per segment; because both users go through the
// functions below (same DPP sequences, same order of the per-ray combination, -ffp-contract=off in every translation
// unit), the fused path and the two-kernel path are BIT-EQUAL (tests/test_gpu_stages.py).
#pragma once
#include <hip/hip_runtime.h>

namespace objnerf {

// Cross-lane steps as DPP modifiers on VALU instructions (no LDS crossbar traffic: a __shfl is a ds_bpermute_b32).
// dpp_ctrl: quad_perm 0x00-0xff, row_shr:n 0x110+n, row_ror:n 0x120+n, wave_shr:1 0x138, row_bcast:15 0x142,
// row_bcast:31 0x143 (gfx9 encodings); lanes without a source (or masked rows / banks) keep `old`.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}
__device__ __forceinline__ float lane_value(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// alpha of one sample (rendering.py:156-157; the caller has already added the noise term to sigma)
__device__ __forceinline__ float sample_alpha(float delta, float sigma) { return 1.f - expf(-delta * fmaxf(sigma, 0.f)); }

// inclusive product scan inside each 32-lane half of the wave: Kogge-Stone inside the rows of 16 (row_shr 1, 2, 4, 8;
// lanes without a source multiply by 1), then the first row's total into the second row of the half (row_bcast:15 -> rows 1, 3)
__device__ __forceinline__ float seg32_scan_mul(float v) {
  v *= dpp<0x111>(1.f, v);
  v *= dpp<0x112>(1.f, v);
  v *= dpp<0x114>(1.f, v);
  v *= dpp<0x118>(1.f, v);
  v *= dpp<0x142, 0xa>(1.f, v);
  return v;
}
// sum over each 32-lane half; valid in the LAST lane of the half (lanes 31 and 63)
__device__ __forceinline__ float seg32_sum_last(float v) {
  v += dpp<0xB1>(0.f, v);            // quad_perm [1,0,3,2]
  v += dpp<0x4E>(0.f, v);            // quad_perm [2,3,0,1]
  v += dpp<0x124>(0.f, v);           // row_ror:4
  v += dpp<0x128>(0.f, v);           // row_ror:8   -> every lane of a row holds the row's sum
  v += dpp<0x142, 0xa>(0.f, v);      // row_bcast:15: rows 1 and 3 add the row in front of them
  return v;
}

struct SegTotals { float Q, A, R, G, B, D; };

// One segment per 32-lane half.  alpha: this lane's sample (0 where `in` is false), c / z: its colour and depth.
// Returns the lane's local weight lw; lo / hi receive the totals of the lower / upper half's segment (wave-uniform).
__device__ __forceinline__ float segment_composite(float alpha, bool in, float c0, float c1, float c2, float z, int lane,
                                                   SegTotals& lo, SegTotals& hi) {
  const float t = in ? (1.f - alpha) + 1e-10f : 1.f;
  const float incl = seg32_scan_mul(t);
  const float below = dpp<0x138>(1.f, incl);                 // wave_shr:1
  const float excl = (lane & 31) == 0 ? 1.f : below;
  const float lw = alpha * excl;
  const float a = seg32_sum_last(lw), r = seg32_sum_last(lw * c0), g = seg32_sum_last(lw * c1), b = seg32_sum_last(lw * c2),
              d = seg32_sum_last(lw * z);
  lo.Q = lane_value(incl, 31); hi.Q = lane_value(incl, 63);
  lo.A = lane_value(a, 31); hi.A = lane_value(a, 63);
  lo.R = lane_value(r, 31); hi.R = lane_value(r, 63);
  lo.G = lane_value(g, 31); hi.G = lane_value(g, 63);
  lo.B = lane_value(b, 31); hi.B = lane_value(b, 63);
  lo.D = lane_value(d, 31); hi.D = lane_value(d, 63);
  return lw;
}

// per-ray combination, one segment at a time (ascending)
struct RayAcc {
  float T = 1.f, opacity = 0.f, r = 0.f, g = 0.f, b = 0.f, depth = 0.f;
  __device__ __forceinline__ float step(const SegTotals& s) {     // returns the segment's incoming transmittance
    const float Tin = T;
    opacity = opacity + Tin * s.A;
    r = r + Tin * s.R; g = g + Tin * s.G; b = b + Tin * s.B;
    depth = depth + Tin * s.D;
    T = Tin * s.Q;
    return Tin;
  }
};

// records the fused MLP epilogue leaves per segment: 64 bytes, [scene Q A R G B D - - | instance Q A R G B D - -]
constexpr int kSegRecFloats = 16;
constexpr int kSegRecInst = 8;

}  // namespace objnerf
