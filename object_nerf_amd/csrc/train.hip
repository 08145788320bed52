// train.hip -- training path of the dual-branch MLP: layer-wise forward that keeps
// the activations, and the matching backward (dgrad + wgrad + bias grads), built on the fp32 MFMA GEMM of
// gemm.h plus a few element-wise kernels.  Operates on the reference's own nn.Linear tensors (row-major
// (out,in) weights, no packing) and on dense row-major activations, i.e. exactly the tensors
// ObjectNeRF.forward / forward_instance see (models/nerf_model.py:97-152); torch.cat of the skip / direction
// inputs is never materialised (column-block GEMMs accumulate into the same output instead).
//
// The inference path (mlp_kernel.h) keeps everything in registers and cannot be differentiated; this path
// trades that for saved activations (12.6 KB per sample point) and is what training_step (train.py:147-180)
// runs through the autograd wrapper in object_nerf_amd/autograd.py.
#include <hip/hip_runtime.h>
#include <string.h>
#include "layout.h"
#include <stdlib.h>
#include "gemm.h"
#include "wgrad.h"
#include "host_api.h"
#include <mutex>
#include <vector>

namespace objnerf {

// ---- measurement hook (bench.py train_step.phases_ms): HIP events at the phase boundaries of the training calls --------
enum { PH_FORWARD = 0, PH_DGRAD, PH_DX, PH_SCATTER, PH_WGRAD, PH_COUNT };
static_assert(PH_COUNT == OBJNERF_TRAIN_PHASES, "phase list of objnerf_train_timing_read");
static std::mutex g_pmu;
static bool g_phase_timing = false;
struct PhaseSpan { int phase; hipEvent_t e0, e1; };
static std::vector<PhaseSpan> g_spans;
// brackets consecutive phases of a call on its stream: begin(p) closes the phase before it; every span owns its two events
struct PhaseClock {
  hipStream_t s;
  bool on;
  int cur = -1;
  hipEvent_t start = nullptr;
  explicit PhaseClock(hipStream_t st) : s(st) { std::lock_guard<std::mutex> lk(g_pmu); on = g_phase_timing; }
  void begin(int phase) {
    if (!on) return;
    if (cur >= 0) {
      hipEvent_t e = nullptr;
      hipEventCreate(&e);
      hipEventRecord(e, s);
      std::lock_guard<std::mutex> lk(g_pmu);
      g_spans.push_back({cur, start, e});
    }
    cur = phase;
    if (phase >= 0) { hipEventCreate(&start); hipEventRecord(start, s); }
  }
  void end() { begin(-1); }
};

template <bool TAIL>
static void gemm_launch_part(const GemmArgs& g, hipStream_t s) {
  dim3 grid(gemm_grid(g.M, (unsigned)g.nbx, g.split_k));
  if (g.a_k_contig && g.b_k_contig) hipLaunchKernelGGL((gemm_kernel<true, true, TAIL>), grid, dim3(256), 0, s, g);
  else if (g.a_k_contig && !g.b_k_contig) hipLaunchKernelGGL((gemm_kernel<true, false, TAIL>), grid, dim3(256), 0, s, g);
  else if (!g.a_k_contig && g.b_k_contig) hipLaunchKernelGGL((gemm_kernel<false, true, TAIL>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_kernel<false, false, TAIL>), grid, dim3(256), 0, s, g);
}

int gemm_launch(const GemmArgs& g0, hipStream_t s) {
  if (g0.M <= 0 || g0.N <= 0) return 0;
  if (g0.K <= 0) return 0;
  GemmArgs g = g0;
  const int nx = (int)((g.N + GBN - 1) / GBN);
  const long last = g.N - (long)(nx - 1) * GBN;              // columns of the last column tile
  const bool tail = OBJ_GEMM_TAIL && last <= 96;             // at most 3 of its four 32-column sub-tiles are live
  g.bx0 = 0;
  g.nbx = tail ? nx - 1 : nx;
  if (g.nbx > 0) gemm_launch_part<false>(g, s);
  if (tail) {
    g.bx0 = nx - 1;
    g.nbx = 1;
    gemm_launch_part<true>(g, s);
  }
  return check_launch("gemm");
}

// ---- element-wise helpers ------------------------------------------------------------------------
__global__ void sigmoid_bwd_kernel(float* __restrict__ dz, const float* __restrict__ dy, const float* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dz[i] = dy[i] * (y[i] * (1.f - y[i]));
}
struct Ctx {
  hipStream_t s;
  int rc;
};

static inline unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }

// C = A' * B' helpers over row-major activations / nn.Linear weights
static void lin_fwd(Ctx& c, const float* X, long ldx, const float* W, long ldw, long P, int out, int in, float* Y, long ldy,
                    int accumulate, int epi, const float* bias) {
  if (c.rc) return;
  GemmArgs g{X, ldx, 1, W, ldw, 1, Y, ldy, P, out, in, accumulate, epi, bias, 1, nullptr};
  c.rc = gemm_launch(g, c.s);
}
// dX (P x in) (+)= dY (P x out) * W (out x in)
// `act` (P x in, leading dimension lddx): the saved output of the LeakyReLU layer dX belongs to; the activation's
// backward is then applied to the complete sum in the epilogue instead of in a pass of its own
static void lin_dgrad(Ctx& c, const float* dY, long lddy, const float* W, long ldw, long P, int out, int in, float* dX,
                      long lddx, int accumulate, const float* act = nullptr) {
  if (c.rc) return;
  GemmArgs g{dY, lddy, 1, W, ldw, 0, dX, lddx, P, in, out, accumulate, act ? EPI_LEAKY_BWD : EPI_NONE, act, 1, nullptr};
  c.rc = gemm_launch(g, c.s);
}
// dX (P x in) = sum_s dY_s (P x out_s) * W_s (out_s x in): the gradient w.r.t. an input that feeds several layers, as ONE
// product over the concatenated contraction range (gemm.h "segmented contraction") instead of one accumulating launch
// per consumer: no read-modify-write passes over dX and a 2-4x longer k loop per output tile
struct DgradSeg { const float* dY; long lddy; const float* W; long ldw; int out; };
static void lin_dgrad_multi(Ctx& c, const DgradSeg* segs, int nseg, long P, int in, float* dX, long lddx) {
  if (c.rc || nseg <= 0) return;
  GemmArgs g{segs[0].dY, segs[0].lddy, 1, segs[0].W, segs[0].ldw, 0, dX, lddx, P, in, segs[0].out, 0, EPI_NONE, nullptr, 1, nullptr};
  if (nseg > 1) {
    g.nseg = nseg;
    for (int i = 0; i < nseg; ++i) {
      g.segA[i] = segs[i].dY; g.seglda[i] = segs[i].lddy;
      g.segB[i] = segs[i].W; g.segldb[i] = segs[i].ldw;
      g.segK[i] = segs[i].out;
    }
  }
  c.rc = gemm_launch(g, c.s);
}
// dW (out x in, ld ldw) += dY^T (out x P) * X (P x in); split over the points
// db (optional): the layer's bias gradient, db[o] += sum_p dY[p, o], taken from the dY^T tiles this GEMM stages anyway
static void lin_wgrad(Ctx& c, const float* dY, long lddy, const float* X, long ldx, long P, int out, int in, float* dW,
                      long ldw, float* db = nullptr) {
  if (c.rc) return;
  // Split over the points so that a workgroup contracts ~1024 of them (~512 when the output is a single tile):
  // measured optimum on MI355X for P = 1.3e5 .. 2.6e5 (tools/split_bench.py) -- fewer splits leave CUs idle, more
  // splits pay the 16K-atomic epilogue per workgroup for too little work.
  const long tiles = ((out + GBM - 1) / GBM) * (long)((in + GBN - 1) / GBN);
  long split = P / (tiles >= 2 ? 1024 : 512);
  const long max_split = (P + 4 * GBK - 1) / (4 * GBK);
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  GemmArgs g{dY, lddy, 0, X, ldx, 0, dW, ldw, out, in, P, 1, EPI_NONE, nullptr, (int)(split > 1 ? split : 2), db};
  c.rc = gemm_launch(g, c.s);       // split_k >= 2 forces the atomic += path (dW always accumulates)
}
// saved-activation layout (floats per point)
struct Ws {
  long P;
  float* base;
  bool obj;
  float* A(int l) const { return base + (long)(l - 1) * 256 * P; }            // scene layers 1..8
  float* final_() const { return base + 8L * 256 * P; }
  float* dirh() const { return final_() + 256L * P; }
  float* rgb() const { return dirh() + 128L * P; }
  float* B(int l) const { return rgb() + 4L * P + (long)(l - 1) * 128 * P; }   // object layers 1..4
  float* ofinal() const { return B(5); }
  float* odirh() const { return ofinal() + 128L * P; }
  float* irgb() const { return odirh() + 64L * P; }
};
constexpr long kWsScene = 8L * 256 + 256 + 128 + 4;
constexpr long kWsObj = 4L * 128 + 128 + 64 + 4;

}  // namespace objnerf

using namespace objnerf;

extern "C" {

int objnerf_gemm(const float* A, int64_t lda, int a_k_contig, const float* B, int64_t ldb, int b_k_contig, float* C,
                 int64_t ldc, int64_t M, int64_t N, int64_t K, int accumulate, int epilogue, const float* bias,
                 int split_k, void* stream) {
  if (!A || !B || !C || split_k < 1) return set_error(-1, "gemm: bad arguments");
  if (split_k > 1 && epilogue != EPI_NONE) return set_error(-1, "gemm: epilogue needs split_k == 1");
  GemmArgs g{A, lda, a_k_contig, B, ldb, b_k_contig, C, ldc, M, N, K, accumulate, epilogue, bias, split_k, nullptr};
  return gemm_launch(g, (hipStream_t)stream);
}

int objnerf_train_timing_enable(int on) {
  std::lock_guard<std::mutex> lk(g_pmu);
  g_phase_timing = on != 0;
  for (auto& sp : g_spans) { hipEventDestroy(sp.e0); hipEventDestroy(sp.e1); }
  g_spans.clear();
  return 0;
}
int objnerf_train_timing_read(double* ms_by_phase, int64_t* spans_by_phase) {
  std::lock_guard<std::mutex> lk(g_pmu);
  double ms[PH_COUNT] = {0};
  int64_t cnt[PH_COUNT] = {0};
  for (auto& sp : g_spans) {
    hipEventSynchronize(sp.e1);
    float t = 0;
    hipEventElapsedTime(&t, sp.e0, sp.e1);
    if (sp.phase >= 0 && sp.phase < PH_COUNT) { ms[sp.phase] += t; ++cnt[sp.phase]; }
    hipEventDestroy(sp.e0); hipEventDestroy(sp.e1);
  }
  g_spans.clear();
  for (int i = 0; i < PH_COUNT; ++i) {
    if (ms_by_phase) ms_by_phase[i] = ms[i];
    if (spans_by_phase) spans_by_phase[i] = cnt[i];
  }
  return 0;
}

// saved activations, then (round 5) the LeakyReLU sign masks the fused forward packs for the fused dgrad chain (mlp_kernel.h:
// 448 B per point, written only by a forward that runs with objnerf_train_args.blob)
static long ws_act_floats(bool do_object, long n_points) { return (kWsScene + (do_object ? kWsObj : 0)) * n_points; }
int64_t objnerf_train_workspace_floats(int do_object, int64_t n_points) {
  return ws_act_floats(do_object != 0, n_points) + train_mask_floats_host(n_points);
}
// gradients w.r.t. every layer's pre-activation output (the activation workspace's layout), then the partial tiles of
// the grouped weight-gradient pass (wgrad.h)
// ... then (round 5) the per-ray terms' area: column sums of four layers' gradients per 16 points, the per-segment input rows
// they are contracted with, and the partial tiles of that second, small weight-gradient pass
constexpr long kSegCols = 128 + 128 + 128 + 64;                 // O1 | O3 | scene direction layer | object direction layer
constexpr long kSegXLd = 92;                                    // code (64) | direction embedding (27) | pad
static long seg_count(long n_points) { return (n_points + 15) / 16; }
static long seg_area_floats(long n_points) {
  const long ns = seg_count(n_points);
  return ns * (kSegCols + kSegXLd) + wgrad_scratch_floats(ns);
}
int64_t objnerf_train_scratch_floats(int64_t n_points) {
  return (kWsScene + kWsObj) * n_points + wgrad_scratch_floats(n_points) + seg_area_floats(n_points);
}

int objnerf_mlp_train_forward(const objnerf_train_args* a, void* stream) {
  if (!a || !a->h_params || !a->emb_xyz || !a->workspace || !a->sigma || !a->rgb)
    return set_error(-1, "mlp_train_forward: bad arguments");
  if (a->do_object && ((a->use_voxel && !a->obj_voxel) || !a->inst_sigma || !a->inst_rgb))
    return set_error(-1, "mlp_train_forward: object branch inputs/outputs missing");
  // the per-point direction embeddings / object codes are read only by a forward that does not embed in registers
  const bool fused_inputs = a->blob && a->rays;
  if (!fused_inputs && (!a->emb_dir || (a->do_object && !a->obj_code)))
    return set_error(-1, "mlp_train_forward: emb_dir / obj_code are required unless the forward embeds in registers (blob + rays)");
  const long P = a->n_points;
  if (P == 0) return 0;
  const bool vox = a->use_voxel != 0;
  const int cx = in_xyz(vox), co = in_obj(vox);
  const float* const* p = a->h_params;
  auto Wt = [&](int id) { return p[2 * id]; };
  auto Bi = [&](int id) { return p[2 * id + 1]; };
  if (a->blob && !a->aux) return set_error(-1, "mlp_train_forward: blob needs aux");
  PhaseClock clk((hipStream_t)stream);
  clk.begin(PH_FORWARD);
  struct EndClock { PhaseClock& c; ~EndClock() { c.end(); } } end_clock{clk};
  if (a->blob) {
    // persistent MFMA kernel (mlp_kernel.h, memory form) that also writes the activation matrices: one launch per
    // branch; same workspace layout (struct Ws here = struct SaveWs there)
    objnerf_mlp_args m;
    memset(&m, 0, sizeof(m));
    m.use_voxel = a->use_voxel;
    m.blob = a->blob; m.aux = a->aux;
    m.emb_xyz = a->emb_xyz; m.emb_dir = a->emb_dir; m.obj_voxel = a->obj_voxel; m.obj_code = a->obj_code;
    m.n_points = P;
    m.sigma = a->sigma; m.rgb = a->rgb; m.inst_sigma = a->inst_sigma; m.inst_rgb = a->inst_rgb;
    const long ntiles = (P + 127) / 128;
    const unsigned grid = mlp_grid(ntiles);
#ifdef OBJ_NO_MASKS          // A/B build switch: round 4's form (no masks written, the chain re-reads the activations)
    unsigned* mask_ws = nullptr;
#else
    unsigned* mask_ws = (unsigned*)(a->workspace + ws_act_floats(a->do_object != 0, P));
#endif
    if (a->rays) {
      // embeddings computed in registers from (rays, z, grid, codes): both branches in one launch
      if (!a->z_vals || a->S < 1 || a->n_rays * (int64_t)a->S != P || (a->do_object && !a->codes))
        return set_error(-1, "mlp_train_forward: bad fused inputs");
      if (a->use_voxel && (!a->grid.idx_map || !a->grid.table || a->grid.n_rows < 1))
        return set_error(-1, "mlp_train_forward: voxel mode needs a voxel grid");
      m.rays = a->rays; m.z_vals = a->z_vals; m.n_rays = a->n_rays; m.S = a->S;
      m.codes = a->codes; m.code_stride = a->code_stride; m.grid = a->grid;
      m.emb_xyz = nullptr; m.emb_dir = nullptr; m.obj_voxel = nullptr; m.obj_code = nullptr;
      m.do_scene = 1; m.do_object = a->do_object ? 1 : 0;
      // ray_bias_ws: the terms that are constant along a ray arrive per ray (objnerf_ray_bias; the compact weight columns it
      // reads sit behind the aux block: objnerf_pack_weights of the SAME parameter values) and their k-steps are skipped,
      // exactly as in the inference passes (objnerf_mlp_args.ray_bias)
      if (a->ray_bias_ws && [] { const char* e = getenv("OBJNERF_HOIST"); return !e || atoi(e) != 0; }()) {
        const int rc = objnerf_ray_bias(&m, a->ray_bias_ws, stream);
        if (rc) return rc;
        m.ray_bias = a->ray_bias_ws;
      }
      return launch_mlp_fused(m, ntiles, grid, (hipStream_t)stream, a->workspace, mask_ws);
    }
    m.do_scene = 1; m.do_object = 0;
    int rc = launch_mlp_memory(m, ntiles, grid, (hipStream_t)stream, a->workspace, mask_ws);
    if (rc == 0 && a->do_object) {
      m.do_scene = 0; m.do_object = 1;
      rc = launch_mlp_memory(m, ntiles, grid, (hipStream_t)stream, a->workspace, mask_ws);
    }
    return rc;
  }
  Ctx c{(hipStream_t)stream, 0};
  Ws w{P, a->workspace, a->do_object != 0};
  const float* X = a->emb_xyz;
  // scene branch, nerf_model.py:97-121
  lin_fwd(c, X, cx, Wt(P_S1), cx, P, 256, cx, w.A(1), 256, 0, EPI_BIAS_LEAKY, Bi(P_S1));
  for (int l = 2; l <= 8; ++l) {
    if (l == 5) {   // cat([input_xyz, h]) -> two column blocks of W5
      lin_fwd(c, X, cx, Wt(P_S5), cx + 256, P, 256, cx, w.A(5), 256, 0, EPI_NONE, nullptr);
      lin_fwd(c, w.A(4), 256, Wt(P_S5) + cx, cx + 256, P, 256, 256, w.A(5), 256, 1, EPI_BIAS_LEAKY, Bi(P_S5));
    } else {
      lin_fwd(c, w.A(l - 1), 256, Wt(P_S1 + l - 1), 256, P, 256, 256, w.A(l), 256, 0, EPI_BIAS_LEAKY, Bi(P_S1 + l - 1));
    }
  }
  lin_fwd(c, w.A(8), 256, Wt(P_SSIG), 256, P, 1, 256, a->sigma, 1, 0, EPI_BIAS, Bi(P_SSIG));
  lin_fwd(c, w.A(8), 256, Wt(P_SF), 256, P, 256, 256, w.final_(), 256, 0, EPI_BIAS, Bi(P_SF));
  lin_fwd(c, w.final_(), 256, Wt(P_SD), 256 + kDirC, P, 128, 256, w.dirh(), 128, 0, EPI_NONE, nullptr);
  lin_fwd(c, a->emb_dir, kDirC, Wt(P_SD) + 256, 256 + kDirC, P, 128, kDirC, w.dirh(), 128, 1, EPI_BIAS_LEAKY, Bi(P_SD));
  lin_fwd(c, w.dirh(), 128, Wt(P_SRGB), 128, P, 3, 128, a->rgb, 3, 0, EPI_BIAS_SIGMOID, Bi(P_SRGB));
  if (a->do_object) {
    // input_x = cat([emb_xyz, obj_voxel, obj_code]) as column blocks (nerf_model.py:128-132)
    auto obj_in = [&](int wid, int ldw, float* out, int epi_last, const float* bias, int accumulate_first) {
      lin_fwd(c, X, cx, Wt(wid), ldw, P, 128, cx, out, 128, accumulate_first, EPI_NONE, nullptr);
      if (vox) lin_fwd(c, a->obj_voxel, kObjVoxPE, Wt(wid) + cx, ldw, P, 128, kObjVoxPE, out, 128, 1, EPI_NONE, nullptr);
      lin_fwd(c, a->obj_code, kCodeC, Wt(wid) + cx + (vox ? kObjVoxPE : 0), ldw, P, 128, kCodeC, out, 128, 1, epi_last, bias);
    };
    obj_in(P_O1, co, w.B(1), EPI_BIAS_LEAKY, Bi(P_O1), 0);
    lin_fwd(c, w.B(1), 128, Wt(P_O2), 128, P, 128, 128, w.B(2), 128, 0, EPI_BIAS_LEAKY, Bi(P_O2));
    lin_fwd(c, w.B(2), 128, Wt(P_O3) + co, co + 128, P, 128, 128, w.B(3), 128, 0, EPI_NONE, nullptr);
    obj_in(P_O3, co + 128, w.B(3), EPI_BIAS_LEAKY, Bi(P_O3), 1);
    lin_fwd(c, w.B(3), 128, Wt(P_O4), 128, P, 128, 128, w.B(4), 128, 0, EPI_BIAS_LEAKY, Bi(P_O4));
    lin_fwd(c, w.B(4), 128, Wt(P_OSIG), 128, P, 1, 128, a->inst_sigma, 1, 0, EPI_BIAS, Bi(P_OSIG));
    lin_fwd(c, w.B(4), 128, Wt(P_OF), 128, P, 128, 128, w.ofinal(), 128, 0, EPI_BIAS, Bi(P_OF));
    lin_fwd(c, w.ofinal(), 128, Wt(P_OD), 128 + kDirC, P, 64, 128, w.odirh(), 64, 0, EPI_NONE, nullptr);
    lin_fwd(c, a->emb_dir, kDirC, Wt(P_OD) + 128, 128 + kDirC, P, 64, kDirC, w.odirh(), 64, 1, EPI_BIAS_LEAKY, Bi(P_OD));
    lin_fwd(c, w.odirh(), 64, Wt(P_ORGB), 64, P, 3, 64, a->inst_rgb, 3, 0, EPI_BIAS_SIGMOID, Bi(P_ORGB));
  }
  return c.rc;
}

int objnerf_mlp_train_backward(const objnerf_train_args* a, const float* d_sigma, const float* d_rgb,
                               const float* d_inst_sigma, const float* d_inst_rgb, float* const* h_param_grads,
                               float* d_emb_xyz, float* d_obj_voxel, float* d_obj_code, float* scratch, void* stream) {
  if (!a || !a->h_params || !h_param_grads || !a->workspace || !scratch || !d_sigma || !d_rgb || !d_emb_xyz)
    return set_error(-1, "mlp_train_backward: bad arguments");
  if (a->do_object && (!d_inst_sigma || !d_inst_rgb || !d_obj_code || (a->use_voxel && !d_obj_voxel)))
    return set_error(-1, "mlp_train_backward: object branch gradients missing");
  // Per-ray terms (round 5).  The direction embedding and the object code are constant along a ray, so the gradients of the
  // weight columns they meet (dir_encoding / inst_dir_encoding: 27 columns, instance_encoding_1 / _3: 64 columns) and the
  // gradient w.r.t. the code are functions of the layer gradients SUMMED over the ray's samples: sum_p dY_p x_ray(p) =
  // sum_ray (sum_s dY) x_ray.  The grouped weight-gradient pass leaves those sums per 16 points beside its bias sums
  // (wgrad.h segsum), and a second, small pass contracts them over P / 16 segments instead of P points: four ragged tiles of
  // the main pass, the (P x 64) code-gradient product and the per-point copies of codes / direction embeddings disappear.
  // Needs whole 16-point segments inside a ray (S % 16 == 0) and the per-ray inputs; otherwise the per-point form below.
  const bool per_ray = a->emb_dir_ray != nullptr && a->S >= 16 && a->S % 16 == 0 && a->n_rays * (int64_t)a->S == a->n_points &&
                       (!a->do_object || (a->codes != nullptr && a->code_stride == kCodeC));
  // (the CALLER chooses the form by passing emb_dir_ray or not -- object_nerf_amd/autograd.py decides once, at forward time, and
  // sizes d_obj_code for it; rounds 4-5 re-read OBJNERF_TRAIN_PER_RAY / OBJNERF_WGRAD here and could disagree with the caller)
  if (!per_ray && (!a->emb_dir || (a->do_object && !a->obj_code)))
    return set_error(-1, "mlp_train_backward: emb_dir / obj_code (per point) or emb_dir_ray / codes / S (per ray) are required");
  if (a->blob_bwd && !a->aux) return set_error(-1, "mlp_train_backward: blob_bwd needs aux");
  const long P = a->n_points;
  if (P == 0) return 0;
  const bool vox = a->use_voxel != 0, obj = a->do_object != 0;
  const int cx = in_xyz(vox), co = in_obj(vox);
  const float* const* p = a->h_params;
  auto Wt = [&](int id) { return p[2 * id]; };
  auto gW = [&](int id) { return h_param_grads[2 * id]; };
  auto gB = [&](int id) { return h_param_grads[2 * id + 1]; };
  Ctx c{(hipStream_t)stream, 0};
  const Ws w{P, a->workspace, obj};       // saved activations
  const Ws d{P, scratch, obj};            // gradients w.r.t. the pre-activation outputs, same layout
  const float* X = a->emb_xyz;
  // d_emb_xyz: only the voxel-feature columns have a consumer (the table scatter); the xyz positional-encoding
  // columns would be the gradient w.r.t. the sample positions, which the reference detaches (rendering.py:307)
  const int ce = vox ? kScnVoxPE : 0;
  float* t2 = d.rgb();                    // (P,3) gradient w.r.t. the rgb heads' pre-sigmoid outputs
  float* t2i = d.irgb();

  PhaseClock clk(c.s);
  struct EndClock { PhaseClock& c; ~EndClock() { c.end(); } } end_clock{clk};
  bool dx_fused = false;                  // the fused chain also formed d_emb_xyz / d_obj_voxel (objnerf_train_args.bwd_dx)
  // ---- phase A: the dgrad chain through the hidden layers -> d.* ----
  clk.begin(PH_DGRAD);
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(nblk(3 * P)), dim3(256), 0, c.s, t2, d_rgb, a->rgb, 3 * P);
  if (obj) hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(nblk(3 * P)), dim3(256), 0, c.s, t2i, d_inst_rgb, a->inst_rgb, 3 * P);
  c.rc = check_launch("sigmoid_bwd");
  if (a->blob_bwd) {
    // one persistent MFMA kernel, gradient tiles stay in registers from layer to layer (mlp_bwd.hip).  `blob` in THIS call says
    // that the forward of this workspace ran with blob too, i.e. on the fused kernel, which left the LeakyReLU sign masks behind
    // the activation matrices; after a layer-by-layer forward (no blob) the chain derives them from the activations instead
    // (OBJNERF_BWD_MASKS=0: that form also after a fused forward -- A/B switch, identical gradients)
#ifdef OBJ_NO_MASKS
    const bool use_masks = false;
#else
    const bool use_masks = a->blob != nullptr && [] { const char* e = getenv("OBJNERF_BWD_MASKS"); return !e || atoi(e) != 0; }();
#endif
    const unsigned* masks = use_masks ? (const unsigned*)(a->workspace + ws_act_floats(obj, P)) : nullptr;
    dx_fused = a->bwd_dx != 0;
    if (dx_fused && !vox) return set_error(-1, "mlp_train_backward: bwd_dx marks a voxel-mode stream");
    if (!c.rc) c.rc = launch_mlp_bwd(a->blob_bwd, a->aux, P, a->workspace, scratch, d_sigma, t2, d_inst_sigma, t2i, obj, masks, dx_fused,
                                     d_emb_xyz, cx, obj ? d_obj_voxel : nullptr, c.s);
  } else {
    if (a->bwd_dx) return set_error(-1, "mlp_train_backward: bwd_dx without blob_bwd");
    // layer by layer: dX = dY W as a GEMM with the LeakyReLU backward in its epilogue
    lin_dgrad(c, t2, 3, Wt(P_SRGB), 128, P, 3, 128, d.dirh(), 128, 0, w.dirh());
    lin_dgrad(c, d.dirh(), 128, Wt(P_SD), 256 + kDirC, P, 128, 256, d.final_(), 256, 0);
    lin_dgrad(c, d.final_(), 256, Wt(P_SF), 256, P, 256, 256, d.A(8), 256, 0);
    lin_dgrad(c, d_sigma, 1, Wt(P_SSIG), 256, P, 1, 256, d.A(8), 256, 1, w.A(8));      // + d sigma * w_sigma, then leaky
    for (int l = 8; l >= 2; --l) {
      const float* Wl = l == 5 ? Wt(P_S5) + cx : Wt(P_S1 + l - 1);                      // hidden block of the skip layer
      lin_dgrad(c, d.A(l), 256, Wl, l == 5 ? cx + 256 : 256, P, 256, 256, d.A(l - 1), 256, 0, w.A(l - 1));
    }
    if (obj) {
      lin_dgrad(c, t2i, 3, Wt(P_ORGB), 64, P, 3, 64, d.odirh(), 64, 0, w.odirh());
      lin_dgrad(c, d.odirh(), 64, Wt(P_OD), 128 + kDirC, P, 64, 128, d.ofinal(), 128, 0);
      lin_dgrad(c, d.ofinal(), 128, Wt(P_OF), 128, P, 128, 128, d.B(4), 128, 0);
      lin_dgrad(c, d_inst_sigma, 1, Wt(P_OSIG), 128, P, 1, 128, d.B(4), 128, 1, w.B(4));
      for (int l = 4; l >= 2; --l) {
        const float* Wl = l == 3 ? Wt(P_O3) + co : Wt(P_O1 + l - 1);
        lin_dgrad(c, d.B(l), 128, Wl, l == 3 ? co + 128 : 128, P, 128, 128, d.B(l - 1), 128, 0, w.B(l - 1));
      }
    }
  }

  // ---- phase B0 (before the weight gradients: the dgrad results are consumed while they are cache-hot): gradients w.r.t.
  // the embeddings: every consumer layer's dY W block in one segmented product per input ----
  clk.begin(PH_DX);
  {
    const int ov = vox ? kObjVoxPE : 0;
    const DgradSeg emb[4] = {{d.A(5), 256, Wt(P_S5), cx + 256, 256}, {d.A(1), 256, Wt(P_S1), cx, 256},
                             {d.B(3), 128, Wt(P_O3), co + 128, 128}, {d.B(1), 128, Wt(P_O1), co, 128}};
    // (dx_fused: the chain kernel has formed both already, from the gradient tiles while they were in registers)
    if (ce && !dx_fused) lin_dgrad_multi(c, emb, obj ? 4 : 2, P, ce, d_emb_xyz, cx);       // only the voxel-feature columns (see above)
    if (obj) {
      const DgradSeg ovs[2] = {{d.B(3), 128, Wt(P_O3) + cx, co + 128, 128}, {d.B(1), 128, Wt(P_O1) + cx, co, 128}};
      if (vox && !dx_fused) lin_dgrad_multi(c, ovs, 2, P, kObjVoxPE, d_obj_voxel, kObjVoxPE);
      const DgradSeg cds[2] = {{d.B(3), 128, Wt(P_O3) + cx + ov, co + 128, 128}, {d.B(1), 128, Wt(P_O1) + cx + ov, co, 128}};
      if (!per_ray) lin_dgrad_multi(c, cds, 2, P, kCodeC, d_obj_code, kCodeC);
    }
  }
  // ---- the voxel-table scatter of those gradients (objnerf_train_args.scatter_*).  On a SIDE stream beside the weight-gradient
  // kernels it was measured slower (20.30 vs 20.05 ms per step, profiles/r04_train_ab.txt: its workgroups take compute units from
  // MFMA-bound kernels), so it is simply enqueued here ----
  clk.begin(PH_SCATTER);
  if (vox && a->scatter_xyz && a->scatter_table_grad && !c.rc)
    c.rc = launch_voxel_embed_bwd(&a->grid, a->scatter_xyz, P, d_emb_xyz, obj ? d_obj_voxel : nullptr, a->scatter_table_grad,
                                  a->emb_xyz, obj ? a->obj_voxel : nullptr, c.s);       // (the rows' identity blocks: the features)

  // ---- phase B: weight / bias gradients dW = dY^T X.  All products of the pass go into ONE work list and one persistent,
  // deterministic stream-K launch (wgrad.h); OBJNERF_WGRAD=atomic keeps round 2's one split-K launch with atomic
  // accumulation per product (developer A/B switch) ----
  // (read on every call, not cached: tests/test_gpu_train.py switches it inside one process to cross-check the two paths)
  const bool atomic_wgrad = [] { const char* e = getenv("OBJNERF_WGRAD"); return e && !strcmp(e, "atomic"); }();
  clk.begin(PH_WGRAD);
  WgradBatch batch;
  // seg (per_ray only): where this layer's 16-point column sums go (column offset inside the segment-sum rows)
  float* const segsum = scratch + (kWsScene + kWsObj) * P + wgrad_scratch_floats(P);       // (P / 16) x kSegCols
  auto wgrad = [&](const float* dY, long lddy, const float* Xo, long ldx, long /*P*/, int out, int in, float* dW, long ldw, float* db = nullptr,
                   int seg = -1) {
    if (atomic_wgrad) { lin_wgrad(c, dY, lddy, Xo, ldx, P, out, in, dW, ldw, db); return; }
    if (out <= 3) batch.add_head(dY, out, Xo, ldx, in, dW, ldw, db);
    else batch.add(dY, lddy, Xo, ldx, out, in, dW, ldw, db, (per_ray && seg >= 0) ? segsum + seg : nullptr, kSegCols);
  };
  // scene heads and the direction layer (cat([final, emb_dir]) as column blocks)
  wgrad(t2, 3, w.dirh(), 128, P, 3, 128, gW(P_SRGB), 128, gB(P_SRGB));
  wgrad(d.dirh(), 128, w.final_(), 256, P, 128, 256, gW(P_SD), 256 + kDirC, gB(P_SD), 256);
  if (!per_ray) wgrad(d.dirh(), 128, a->emb_dir, kDirC, P, 128, kDirC, gW(P_SD) + 256, 256 + kDirC);
  wgrad(d.final_(), 256, w.A(8), 256, P, 256, 256, gW(P_SF), 256, gB(P_SF));
  wgrad(d_sigma, 1, w.A(8), 256, P, 1, 256, gW(P_SSIG), 256, gB(P_SSIG));
  for (int l = 8; l >= 1; --l) {
    float* db = gB(P_S1 + l - 1);
    if (l == 5) {          // cat([input_xyz, h])
      wgrad(d.A(5), 256, X, cx, P, 256, cx, gW(P_S5), cx + 256, db);
      wgrad(d.A(5), 256, w.A(4), 256, P, 256, 256, gW(P_S5) + cx, cx + 256);
    } else if (l == 1) {
      wgrad(d.A(1), 256, X, cx, P, 256, cx, gW(P_S1), cx, db);
    } else {
      wgrad(d.A(l), 256, w.A(l - 1), 256, P, 256, 256, gW(P_S1 + l - 1), 256, db);
    }
  }
  if (obj) {
    const int ov = vox ? kObjVoxPE : 0;
    wgrad(t2i, 3, w.odirh(), 64, P, 3, 64, gW(P_ORGB), 64, gB(P_ORGB));
    wgrad(d.odirh(), 64, w.ofinal(), 128, P, 64, 128, gW(P_OD), 128 + kDirC, gB(P_OD), 384);
    if (!per_ray) wgrad(d.odirh(), 64, a->emb_dir, kDirC, P, 64, kDirC, gW(P_OD) + 128, 128 + kDirC);
    wgrad(d.ofinal(), 128, w.B(4), 128, P, 128, 128, gW(P_OF), 128, gB(P_OF));
    wgrad(d_inst_sigma, 1, w.B(4), 128, P, 1, 128, gW(P_OSIG), 128, gB(P_OSIG));
    // one layer fed by cat([emb_xyz, obj_voxel, obj_code]): three column blocks of its weight
    auto obj_in_bwd = [&](const float* dY, int wid, int ldw, float* db, int seg) {
      wgrad(dY, 128, X, cx, P, 128, cx, gW(wid), ldw, db, seg);
      if (vox) wgrad(dY, 128, a->obj_voxel, kObjVoxPE, P, 128, kObjVoxPE, gW(wid) + cx, ldw);
      if (!per_ray) wgrad(dY, 128, a->obj_code, kCodeC, P, 128, kCodeC, gW(wid) + cx + ov, ldw);
    };
    for (int l = 4; l >= 1; --l) {
      float* db = gB(P_O1 + l - 1);
      if (l == 3) {        // cat([input_x, x_])
        wgrad(d.B(3), 128, w.B(2), 128, P, 128, 128, gW(P_O3) + co, co + 128);
        obj_in_bwd(d.B(3), P_O3, co + 128, db, 128);
      } else if (l == 1) {
        obj_in_bwd(d.B(1), P_O1, co, db, 0);
      } else {
        wgrad(d.B(l), 128, w.B(l - 1), 128, P, 128, 128, gW(P_O1 + l - 1), 128, db);
      }
    }
  }
  if (!atomic_wgrad && !c.rc) c.rc = batch.launch(P, scratch + (kWsScene + kWsObj) * P, c.s);
  if (per_ray && !c.rc) {
    // ---- the per-ray terms from the segment sums the pass above left: K = P / 16 segments ----
    const long ns = seg_count(P);
    const int rep = a->S / 16;                                    // segments per ray
    float* xseg = segsum + ns * kSegCols;                         // ns x kSegXLd: [code | direction embedding] of the segment's ray
    float* scratch2 = xseg + ns * kSegXLd;
    c.rc = objnerf_repeat_rows(a->emb_dir_ray, kDirC, ns, kDirC, rep, xseg + kCodeC, kSegXLd, c.s);       // (ns OUTPUT rows)
    if (!c.rc && obj) c.rc = objnerf_repeat_rows(a->codes, a->code_stride, ns, kCodeC, rep, xseg, kSegXLd, c.s);
    WgradBatch b2;
    b2.add(segsum + 256, kSegCols, xseg + kCodeC, kSegXLd, 128, kDirC, gW(P_SD) + 256, 256 + kDirC, nullptr);
    if (obj) {
      const int ov = vox ? kObjVoxPE : 0;
      b2.add(segsum + 384, kSegCols, xseg + kCodeC, kSegXLd, 64, kDirC, gW(P_OD) + 128, 128 + kDirC, nullptr);
      b2.add(segsum + 0, kSegCols, xseg, kSegXLd, 128, kCodeC, gW(P_O1) + cx + ov, co, nullptr);
      b2.add(segsum + 128, kSegCols, xseg, kSegXLd, 128, kCodeC, gW(P_O3) + cx + ov, co + 128, nullptr);
      // gradient w.r.t. the code, per segment (the caller sums a ray's S / 16 segments): d_obj_code is (P / 16, 64) here
      const DgradSeg cds[2] = {{segsum + 128, kSegCols, Wt(P_O3) + cx + ov, co + 128, 128}, {segsum + 0, kSegCols, Wt(P_O1) + cx + ov, co, 128}};
      lin_dgrad_multi(c, cds, 2, ns, kCodeC, d_obj_code, kCodeC);
    }
    // four ragged tiles contracting over P / 16 rows: slices of 8 k tiles (256 segments) give the device ~100 workgroups instead
    // of ~16; the slot area (sized for 64 tiles x the default slice count of ns rows) holds 4 tiles x 16 times as many slices
    const long kt2 = (ns + GBK - 1) / GBK;
    long sl2 = (kt2 + 7) / 8;
    const long cap2 = 16L * wgrad_slices(ns);
    if (sl2 > cap2) sl2 = cap2;
    if (sl2 > kWgradMaxSlices) sl2 = kWgradMaxSlices;
    if (!c.rc) c.rc = b2.launch(ns, scratch2, c.s, (int)(sl2 < 1 ? 1 : sl2));
  }
  return c.rc;
}

}  // extern "C"
