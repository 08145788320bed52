// api.hip -- host side of the C ABI: error state, the weight-packing index maps, the MLP
// dispatcher, the whole-render_rays driver and the measurement hooks.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "layout.h"
#include "host_api.h"

namespace objnerf {

static thread_local char g_err[512] = "";

int set_error(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return -2;
}

// ---- reference parameter shapes (models/nerf_model.py:41-58, 77-95) --------------------------
struct ParamShape { int out, in; };
static ParamShape param_shape(bool voxel, int p) {
  switch (p) {
    case P_S1: return {kW, in_xyz(voxel)};
    case P_S5: return {kW, in_xyz(voxel) + kW};
    case P_S2: case P_S3: case P_S4: case P_S6: case P_S7: case P_S8: case P_SF: return {kW, kW};
    case P_SD: return {kW / 2, kW + kDirC};
    case P_SSIG: return {1, kW};
    case P_SRGB: return {3, kW / 2};
    case P_O1: return {kIW, in_obj(voxel)};
    case P_O3: return {kIW, in_obj(voxel) + kIW};
    case P_O2: case P_O4: case P_OF: return {kIW, kIW};
    case P_OD: return {kIW / 2, kIW + kDirC};
    case P_OSIG: return {1, kIW};
    case P_ORGB: return {3, kIW / 2};
  }
  return {0, 0};
}
static int layer_param(int l) {
  static const int map[L_COUNT] = {P_S1, P_S2, P_S3, P_S4, P_S5, P_S6, P_S7, P_S8, P_SF, P_SD,
                                   P_O1, P_O2, P_O3, P_O4, P_OF, P_OD};
  return map[l];
}
static inline uint32_t enc(int ptr_id, long off) { return ((uint32_t)ptr_id << 24) | (uint32_t)off; }

// ---- measurement hooks ------------------------------------------------------------------------
static std::mutex g_tmu;
static bool g_timing = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_events;

}  // namespace objnerf

using namespace objnerf;

namespace objnerf {
unsigned mlp_grid(long ntiles) {
  int dev = 0, cus = 256;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return (unsigned)(ntiles < cus ? ntiles : cus);   // persistent: 1 workgroup per CU
}
}  // namespace objnerf

extern "C" {

int objnerf_abi_version(void) { return OBJNERF_ABI_VERSION; }
const char* objnerf_last_error(void) { return g_err; }

int64_t objnerf_blob_floats(int use_voxel) { return (int64_t)total_chunks(use_voxel != 0) * kChunkFloats; }
int64_t objnerf_aux_floats(void) { return kAuxFloats + kRbMatFloats; }
int objnerf_num_param_ptrs(void) { return kNumParamPtrs; }
int64_t objnerf_param_numel(int use_voxel, int ptr_id) {
  if (ptr_id < 0 || ptr_id >= kNumParamPtrs) return -1;
  const ParamShape s = param_shape(use_voxel != 0, ptr_id >> 1);
  return (ptr_id & 1) ? s.out : (int64_t)s.out * s.in;
}

int objnerf_pack_index(int use_voxel, uint32_t* blob_idx, uint32_t* aux_idx) {
  if (!blob_idx || !aux_idx) return set_error(-1, "pack_index: null output");
  const bool vox = use_voxel != 0;
  const long nblob = objnerf_blob_floats(use_voxel);
  for (long i = 0; i < nblob; ++i) blob_idx[i] = kPackZero;
  for (int i = 0; i < kAuxFloats + kRbMatFloats; ++i) aux_idx[i] = kPackZero;      // the tail is written by the packer itself

  for (int l = 0; l < L_COUNT; ++l) {
    const int nt = layer_nt(l), kg = kChunkTiles / nt, ks_n = layer_ks(vox, l);
    const int p = layer_param(l);
    const ParamShape sh = param_shape(vox, p);
    if (sh.out != layer_out(l) || sh.in != layer_in(vox, l)) return set_error(-3, "pack_index: layout self-check failed");
    const long base = (long)layer_chunk_start(vox, l) * kChunkFloats;
    for (int ks = 0; ks < ks_n; ++ks) {
      const int chunk = ks / kg, kl = ks % kg, g4 = kl / 4, j = kl % 4;
      for (int m = 0; m < nt; ++m)
        for (int lane = 0; lane < 64; ++lane) {
          const int row = 32 * m + (lane & 31);
          const int col = layer_kcol(vox, l, ks, lane >> 5);
          if (row >= sh.out || col < 0) continue;
          if (col >= sh.in) return set_error(-3, "pack_index: column out of range");
          const long o = base + (long)chunk * kChunkFloats + ((long)(g4 * nt + m) * 64 + lane) * 4 + j;
          blob_idx[o] = enc(2 * p, (long)row * sh.in + col);
        }
    }
    // bias: [m][half][16]
    for (int m = 0; m < nt; ++m)
      for (int h = 0; h < 2; ++h)
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * m + hid_feat(r, h);
          if (row < sh.out) aux_idx[aux_bias_off(l) + (m * 2 + h) * 16 + r] = enc(2 * p + 1, row);
        }
  }
  // heads: [t][half][16] per output row, then the biases
  struct Head { int off, param, nt, rows; };
  const Head heads[4] = {{kAuxSSig, P_SSIG, 8, 1}, {kAuxSRgb, P_SRGB, 4, 3}, {kAuxOSig, P_OSIG, 4, 1}, {kAuxORgb, P_ORGB, 2, 3}};
  for (const Head& hd : heads) {
    const int in = hd.nt * 32;
    for (int c = 0; c < hd.rows; ++c)
      for (int t = 0; t < hd.nt; ++t)
        for (int h = 0; h < 2; ++h)
          for (int r = 0; r < 16; ++r)
            aux_idx[hd.off + c * in + (t * 2 + h) * 16 + r] = enc(2 * hd.param, (long)c * in + 32 * t + hid_feat(r, h));
    for (int c = 0; c < hd.rows; ++c) aux_idx[hd.off + hd.rows * in + c] = enc(2 * hd.param + 1, c);
  }
  return 0;
}

int64_t objnerf_bwd_blob_floats(void) { return (int64_t)bwd_total_chunks(true) * kChunkFloats; }       // (the voxel-mode stream: the longer one)

int objnerf_pack_index_bwd(int use_voxel, uint32_t* blob_idx) {
  if (!blob_idx) return set_error(-1, "pack_index_bwd: null output");
  if (use_voxel < 0 || use_voxel > 2) return set_error(-1, "pack_index_bwd: mode 0 (plain), 1 (voxel) or 2 (voxel + embedding-gradient blocks)");
  const bool vox = use_voxel != 0;
  const bool dx = use_voxel == 2;      // the embedding-gradient blocks ride in the stream (layout.h, BL_X*; objnerf_train_args.bwd_dx)
  const long nblob = objnerf_bwd_blob_floats();
  for (long i = 0; i < nblob; ++i) blob_idx[i] = kPackZero;
  for (int l = 0; l < BL_COUNT; ++l) {
    if (bwd_is_x(l) && !dx) continue;
    const int nt = bwd_nt(l), kg = chunk_ksteps(nt), ks_n = bwd_ks(l);
    const int p = bwd_param(l);
    const ParamShape sh = param_shape(vox, p);
    const int col0 = bwd_col0(vox, l), rows = bwd_rows(l);
    if (2 * ks_n != sh.out || col0 + rows > sh.in || rows > 32 * nt) return set_error(-3, "pack_index_bwd: layout self-check failed");
    const long base = (long)bwd_chunk_start(dx, l) * kChunkFloats;
    for (int ks = 0; ks < ks_n; ++ks) {
      const int chunk = ks / kg, kl = ks % kg, g4 = kl / 4, j = kl % 4;
      for (int m = 0; m < nt; ++m)
        for (int lane = 0; lane < 64; ++lane) {
          const int r = 32 * m + (lane & 31);
          if (r >= rows) continue;                               // padding rows of the last tile stay zero
          const int in_feat = col0 + r;                          // tile row: a feature of the layer's input block
          const int out_feat = hid_feat(ks, lane >> 5);          // k: the feature whose gradient the lane half holds
          const long o = base + (long)chunk * kChunkFloats + ((long)(g4 * nt + m) * 64 + lane) * 4 + j;
          blob_idx[o] = enc(2 * p, (long)out_feat * sh.in + in_feat);
        }
    }
  }
  return 0;
}

int objnerf_mlp_eval(const objnerf_mlp_args* a, void* stream) {
  if (!a || !a->blob || !a->aux) return set_error(-1, "mlp_eval: null weights");
  const bool fused = a->emb_xyz == nullptr;
  // the density query on points / a lattice (tools/extract_mesh.py:63-113): fused form + sigma_only
  const bool query = fused && a->sigma_only;
  if ((fused && !query ? a->n_rays * (int64_t)a->S : a->n_points) == 0) return 0;   // nothing to do
  if (!a->do_scene && !a->do_object) return set_error(-1, "mlp_eval: no branch selected");
  const bool comp = a->comp_w != nullptr;      // compositing in the epilogue: sigma / rgb need not be written
  if (a->ray_bias && (!fused || (a->sigma_only && (a->do_scene || !a->do_object))))
    return set_error(-1, "mlp_eval: ray_bias needs the fused form (with sigma_only: the object query, one vector for all points)");
  if (comp) {
    if (!fused || !a->do_scene || !a->comp_rec || a->ray_index || a->sigma_only || a->S < 32 || (a->S & 31))
      return set_error(-1, "mlp_eval: comp_w needs the fused form, the scene branch, comp_rec, S % 32 == 0 and no ray subset");
  } else {
    if (a->do_scene && !a->sigma) return set_error(-1, "mlp_eval: scene branch needs a sigma output");
    if (a->do_object && !a->inst_sigma) return set_error(-1, "mlp_eval: object branch needs an inst_sigma output");
  }
  // ray subset (objnerf_hip.h: fused form only, both or neither): a list without its count would make the kernel walk all
  // n_rays slots of a list whose tail is uninitialised; a count without a list would silently evaluate the first rays
  if ((a->ray_index == nullptr) != (a->n_active == nullptr))
    return set_error(-1, "mlp_eval: ray_index and n_active go together (both or neither)");
  if (a->ray_index && (!fused || query)) return set_error(-1, "mlp_eval: ray_index / n_active need the fused form (rays + z_vals)");
  long P;
  if (query) {
    if (a->do_scene && a->do_object) return set_error(-1, "mlp_eval(points, sigma_only): one branch per call (contiguous stream window)");
    if (a->n_points < 0) return set_error(-1, "mlp_eval(points, sigma_only): negative n_points");
    if (!a->points) {
      if (!a->lat_x || !a->lat_y || !a->lat_z) return set_error(-1, "mlp_eval(sigma_only, fused): needs points or the three lattice axes");
      if (a->lat_n[0] < 1 || a->lat_n[1] < 1 || a->lat_n[2] < 1 || (int64_t)a->lat_n[0] * a->lat_n[1] * a->lat_n[2] != a->n_points)
        return set_error(-1, "mlp_eval(lattice): n_points must equal lat_n[0] * lat_n[1] * lat_n[2]");
    }
    if (a->do_object && !a->codes) return set_error(-1, "mlp_eval(points, sigma_only): the object query needs its code");
    if (a->use_voxel && (!a->grid.idx_map || !a->grid.table || a->grid.n_rows < 1))
      return set_error(-1, "mlp_eval: voxel mode needs a voxel grid");
    P = a->n_points;
  } else if (fused) {
    if (!a->rays || !a->z_vals || a->S < 1 || a->n_rays < 0) return set_error(-1, "mlp_eval: bad fused inputs");
    if (a->do_object && !a->codes) return set_error(-1, "mlp_eval: object branch needs codes");
    if (a->use_voxel && (!a->grid.idx_map || !a->grid.table || a->grid.n_rows < 1))
      return set_error(-1, "mlp_eval: voxel mode needs a voxel grid");
    P = (long)a->n_rays * a->S;
  } else {
    if (!a->emb_dir || a->n_points < 0) return set_error(-1, "mlp_eval: bad memory-form inputs");
    if (a->do_object && (!a->obj_code || (a->use_voxel && !a->obj_voxel)))
      return set_error(-1, "mlp_eval: forward_instance needs obj_code (and obj_voxel in voxel mode)");
    P = a->n_points;
  }
  if (P == 0) return 0;
  const long ntiles = (P + 127) / 128;
  const unsigned grid = mlp_grid(ntiles);
  hipStream_t s = (hipStream_t)stream;

  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool timing;
  { std::lock_guard<std::mutex> lk(g_tmu); timing = g_timing; }
  if (timing) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, s); }
  const int rc = fused ? launch_mlp_fused(*a, ntiles, grid, s) : launch_mlp_memory(*a, ntiles, grid, s);
  if (timing) {
    hipEventRecord(e1, s);
    std::lock_guard<std::mutex> lk(g_tmu);
    g_events.emplace_back(e0, e1);
  }
  return rc;
}

int objnerf_timing_enable(int on) {
  std::lock_guard<std::mutex> lk(g_tmu);
  g_timing = on != 0;
  for (auto& ev : g_events) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
  g_events.clear();
  return 0;
}
int objnerf_timing_read(int64_t* launches, double* total_ms) {
  std::lock_guard<std::mutex> lk(g_tmu);
  double tot = 0;
  for (auto& ev : g_events) {
    hipEventSynchronize(ev.second);
    float ms = 0;
    hipEventElapsedTime(&ms, ev.first, ev.second);
    tot += ms;
    hipEventDestroy(ev.first); hipEventDestroy(ev.second);
  }
  if (launches) *launches = (int64_t)g_events.size();
  if (total_ms) *total_ms = tot;
  g_events.clear();
  return 0;
}

// ---- whole render_rays (models/rendering.py:233-337) --------------------------------------------
// A pass composites in the MLP kernel's epilogue (objnerf_mlp_args.comp_*) when nothing between sigma and alpha needs
// another ray-wide quantity: no occlusion mask (eval mode or frustum_bound_th <= 0, rendering.py:192), no noise
// (noise_std == 0), and 32-sample segments that tile the ray (S % 32 == 0).
static bool pass_fuses(const objnerf_render_cfg* cfg, int S) {
  const bool occlusion = !cfg->is_eval && cfg->frustum_bound_th > 0.f;
  return !cfg->separate_composite && !occlusion && cfg->noise_std == 0.f && S >= 32 && (S & 31) == 0;
}
// workspace of one pass: fused -> segment records only (64 B per 32 samples); two-kernel form ->
// sigma (N*S) | rgb (3*N*S) | inst_sigma (N*S) | inst_rgb (3*N*S)
static int64_t pass_floats(const objnerf_render_cfg* cfg, int64_t n_rays, int S) {
  return pass_fuses(cfg, S) ? n_rays * (S / 32) * OBJNERF_SEG_REC_FLOATS : n_rays * S * 8;
}
// A pass walks the batch in SLABS of at most kRenderSlabRays rays (MLP kernel -> compositing per slab, back to back on the
// stream): rays are independent, so results are identical, and the workspace -- segment records or sigma / rgb, and the
// per-ray vectors of the hoisted terms -- is bounded by the slab instead of growing with the frame (a 4K frame would
// otherwise carry 15 GB of per-ray vectors).  A 640 x 480 frame is one slab.
constexpr int64_t kRenderSlabRays = 1 << 20;
static bool hoists(const objnerf_render_cfg* cfg) { return !cfg->no_hoist; }
static int64_t slab_rays(int64_t n_rays) { return n_rays < kRenderSlabRays ? n_rays : kRenderSlabRays; }
static int64_t pass_area_floats(const objnerf_render_cfg* cfg, int64_t slab) {
  const int64_t c = pass_floats(cfg, slab, cfg->N_samples);
  const int64_t f = cfg->N_importance > 0 ? pass_floats(cfg, slab, cfg->N_samples + cfg->N_importance) : 0;
  return c > f ? c : f;
}
int64_t objnerf_render_workspace_bytes(const objnerf_render_cfg* cfg, int64_t n_rays) {
  if (!cfg || n_rays < 0) return -1;
  const int64_t slab = slab_rays(n_rays);
  return (int64_t)sizeof(float) * (pass_area_floats(cfg, slab) + (hoists(cfg) ? objnerf_ray_bias_floats(slab) : 0)) + 256;
}

static int render_pass(const objnerf_render_cfg* cfg, const objnerf_render_in* in, const objnerf_render_out* out,
                       const float* blob, const float* aux, int S, const float* noise, const float* noise_inst,
                       void* stream) {
  float* ws = (float*)in->workspace;
  const bool fuse = pass_fuses(cfg, S);
  const bool inst_weights = cfg->rays_in_bbox && cfg->forward_instance;          // rendering.py:228-229
  const int64_t slab = slab_rays(in->n_rays);
  float* rb = ws + pass_area_floats(cfg, slab);
  for (int64_t lo = 0; lo < in->n_rays; lo += slab) {
    const int64_t N = in->n_rays - lo < slab ? in->n_rays - lo : slab;
    float* sigma = ws;
    float* rgb = sigma + N * S;
    float* isig = rgb + 3 * N * S;
    float* irgb = isig + N * S;
    float* z = out->z_vals + lo * S;
    float* w = out->weights + lo * S;

    objnerf_mlp_args m;
    memset(&m, 0, sizeof(m));
    m.use_voxel = cfg->use_voxel; m.do_scene = 1; m.do_object = cfg->forward_instance;
    m.blob = blob; m.aux = aux;
    m.rays = in->rays + lo * 8; m.z_vals = z; m.n_rays = N; m.S = S;
    m.codes = in->codes + lo * in->code_stride; m.code_stride = in->code_stride; m.grid = in->grid;
    if (hoists(cfg)) {
      const int rc = objnerf_ray_bias(&m, rb, stream);
      if (rc) return rc;
      m.ray_bias = rb;
    }
    if (fuse) {
      m.comp_w = w; m.comp_rec = ws;
      m.comp_last_delta = cfg->use_zero_as_last_delta ? 0.f : 1e10f;             // rendering.py:143-153
      m.comp_inst_weights = inst_weights;
      int rc = objnerf_mlp_eval(&m, stream);
      if (rc) return rc;
      rc = objnerf_composite_finish(ws, N, S, cfg->forward_instance, inst_weights, cfg->white_back, w, out->opacity + lo,
                                    out->rgb + lo * 3, out->depth + lo, out->rgb_instance ? out->rgb_instance + lo * 3 : nullptr,
                                    out->depth_instance ? out->depth_instance + lo : nullptr,
                                    out->opacity_instance ? out->opacity_instance + lo : nullptr, stream);
      if (rc) return rc;
      continue;
    }
    m.sigma = sigma; m.rgb = rgb;
    m.inst_sigma = cfg->forward_instance ? isig : nullptr;
    m.inst_rgb = cfg->forward_instance ? irgb : nullptr;
    int rc = objnerf_mlp_eval(&m, stream);
    if (rc) return rc;

    objnerf_composite_args c;
    memset(&c, 0, sizeof(c));
    c.n_rays = N; c.S = S; c.z_vals = z; c.sigma = sigma; c.rgb = rgb;
    c.inst_sigma = m.inst_sigma; c.inst_rgb = m.inst_rgb;
    c.noise = noise ? noise + lo * S : nullptr; c.noise_inst = noise_inst ? noise_inst + lo * S : nullptr;
    c.noise_std = cfg->noise_std;
    c.white_back = cfg->white_back; c.use_zero_as_last_delta = cfg->use_zero_as_last_delta;
    c.occlusion = (!cfg->is_eval && cfg->frustum_bound_th > 0.f) ? 1 : 0;     // rendering.py:192
    c.frustum_bound_th = cfg->frustum_bound_th;
    c.pass_through_mask = in->pass_through_mask ? in->pass_through_mask + lo : nullptr;
    c.rays_in_bbox = inst_weights;
    c.weights = w; c.opacity = out->opacity + lo; c.rgb_map = out->rgb + lo * 3; c.depth = out->depth + lo;
    c.rgb_inst = out->rgb_instance ? out->rgb_instance + lo * 3 : nullptr;
    c.depth_inst = out->depth_instance ? out->depth_instance + lo : nullptr;
    c.opacity_inst = out->opacity_instance ? out->opacity_instance + lo : nullptr;
    rc = objnerf_composite(&c, stream);
    if (rc) return rc;
  }
  return 0;
}

int objnerf_render_rays(const objnerf_render_cfg* cfg, const objnerf_render_in* in, const objnerf_render_out* coarse,
                        const objnerf_render_out* fine, void* stream) {
  if (!cfg || !in || !coarse) return set_error(-1, "render_rays: null argument");
  if (in->n_rays == 0) return 0;          // empty batch: nothing to enqueue (zero-size tensors have null pointers)
  if (!in->rays || !in->workspace || !in->blob_coarse || !in->aux_coarse || !in->z_steps)
    return set_error(-1, "render_rays: missing input");
  if (!in->codes) return set_error(-1, "render_rays: embedding_instance is mandatory (rendering.py:94)");
  if (cfg->N_importance > 0 && (!fine || !in->blob_fine || !in->aux_fine))
    return set_error(-1, "render_rays: N_importance > 0 needs the fine model and outputs");
  if (cfg->N_importance > 0 && cfg->perturb == 0.f && !in->u_det) return set_error(-1, "render_rays: missing u_det");
  if (cfg->N_importance > 0 && cfg->perturb != 0.f && !in->u_rand) return set_error(-1, "render_rays: missing u_rand");
  if (in->n_rays == 0) return 0;
  const int S = cfg->N_samples, I = cfg->N_importance;

  int rc = objnerf_sample_coarse(in->rays, in->z_steps, in->perturb_rand, cfg->perturb, cfg->use_disp, in->n_rays, S,
                                 coarse->z_vals, stream);
  if (rc) return rc;
  rc = render_pass(cfg, in, coarse, in->blob_coarse, in->aux_coarse, S, in->noise[0], in->noise[1], stream);
  if (rc || I <= 0) return rc;
  // sample_pdf(z_mid, weights_coarse[:,1:-1], I, det=(perturb==0)) + sort(cat) (rendering.py:301-313)
  const bool det = cfg->perturb == 0.f;
  rc = objnerf_sample_pdf_merge(coarse->z_vals, coarse->weights, det ? in->u_det : in->u_rand, det ? 0 : I,
                                in->n_rays, S, I, 1e-5f, nullptr, fine->z_vals, stream);
  if (rc) return rc;
  return render_pass(cfg, in, fine, in->blob_fine, in->aux_fine, S + I, in->noise[2], in->noise[3], stream);
}

// ---- whole render_rays_multi (render_tools/multi_rendering.py:160-325) in one enqueue ---------------------------
// workspace per ray set k: zc (N*S) | zf (N*(S+I)) | sigma (N*Smax) | rgb (3*N*Smax) | own weights (N*S) | ray_index (N int32)
// then: n_active (K int32, 256-byte slots) | compaction scratch | staging area of the joint compositing (only when K*(S+I)
// samples of 28 bytes exceed its LDS limit)
namespace {
struct MultiWs {
  int64_t N; int S, I, Smax;
  char* base;
  bool hoist;
  // per set: the arrays listed above, padded to 64 bytes, then (fp32 passes) the per-ray vectors of objnerf_ray_bias
  int64_t head_floats() const { return (N * ((int64_t)S + (S + I) + 4LL * Smax + S) + N + 15) / 16 * 16; }
  int64_t set_floats() const { return head_floats() + (hoist ? (objnerf_ray_bias_floats(N) + 15) / 16 * 16 : 0); }
  float* ray_bias(int k) const { return set(k) + head_floats(); }
  float* set(int k) const { return (float*)base + set_floats() * k; }
  float* zc(int k) const { return set(k); }
  float* zf(int k) const { return zc(k) + N * S; }
  float* sigma(int k) const { return zf(k) + N * (S + I); }
  float* rgb(int k) const { return sigma(k) + N * Smax; }
  float* own(int k) const { return rgb(k) + 3 * N * Smax; }
  int32_t* idx(int k) const { return (int32_t*)(own(k) + N * S); }
  int32_t* count(int K, int k) const { return (int32_t*)((float*)base + set_floats() * K) + 64 * k; }
  int32_t* scratch(int K) const { return count(K, K); }
  float* stage(int K) const { return (float*)(scratch(K) + (objnerf_compact_scratch_ints(N) + 63) / 64 * 64); }
};
constexpr int kMultiMaxSets = 64;
}  // namespace

int64_t objnerf_render_multi_workspace_bytes(const objnerf_render_multi_cfg* cfg, int32_t K, int64_t n_rays) {
  if (!cfg || K < 1 || n_rays < 0) return -1;
  const int I = cfg->N_importance > 0 ? cfg->N_importance : 0;
  MultiWs w{n_rays, cfg->N_samples, I, cfg->N_samples + I, nullptr, !cfg->no_hoist};
  return 4 * (w.set_floats() * K + 64LL * K + (objnerf_compact_scratch_ints(n_rays) + 63) / 64 * 64) +
         objnerf_composite_multi_scratch_bytes(K, cfg->N_samples + I) + 256;
}

int objnerf_render_rays_multi(const objnerf_render_multi_cfg* cfg, const objnerf_render_multi_in* in,
                              const objnerf_render_multi_out* coarse, const objnerf_render_multi_out* fine, void* stream) {
  if (!cfg || !in || !coarse) return set_error(-1, "render_rays_multi: null argument");
  const int K = in->K, S = cfg->N_samples, I = cfg->N_importance > 0 ? cfg->N_importance : 0;
  const int64_t N = in->n_rays;
  if (K < 1 || K > kMultiMaxSets || S < 1 || N < 0) return set_error(-1, "render_rays_multi: bad sizes (1 <= K <= 64)");
  // limits of the stages further down the enqueue, checked BEFORE the first launch (the importance sampler needs
  // S - 2 >= 1 interior bins and stages a ray's bins in LDS; the joint compositing has no sample limit: it stages in LDS up
  // to K*(S+I) = 5,558 samples and in the workspace beyond)
  if (I > 0 && S < 3) return set_error(-1, "render_rays_multi: N_importance > 0 needs N_samples >= 3");
  if (I > 0 && (S > 1025 || S + I > 2048))
    return set_error(-1, "render_rays_multi: the importance sampler takes N_samples <= 1025 and N_samples + N_importance <= 2048 per ray set");
  if (N == 0) return 0;
  if (!in->h_rays || !in->h_obj_ids || !in->workspace || !in->blob_coarse || !in->aux_coarse || !in->z_steps)
    return set_error(-1, "render_rays_multi: missing input");
  if (I > 0 && (!fine || !in->blob_fine || !in->aux_fine)) return set_error(-1, "render_rays_multi: N_importance > 0 needs the fine model and outputs");
  const bool det = cfg->perturb == 0.f;
  if (I > 0 && (det ? !in->u_det : !in->u_rand)) return set_error(-1, "render_rays_multi: missing u_det / u_rand");
  if (cfg->noise_std != 0.f && (!in->noise_coarse || (I > 0 && !in->noise_fine)))
    return set_error(-1, "render_rays_multi: noise_std != 0 needs noise draws");
  for (int k = 0; k < K; ++k) {
    if (!in->h_rays[k]) return set_error(-1, "render_rays_multi: null ray set");
    if (in->h_obj_ids[k] > 0 && !in->code_table) return set_error(-1, "render_rays_multi: object sets need the code table");
    if (in->h_obj_ids[k] < 0) return set_error(-1, "render_rays_multi: negative object id");
  }
  MultiWs w{N, S, I, S + I, (char*)in->workspace, !cfg->no_hoist};

  auto one_pass = [&](bool is_fine, const float* blob, const float* aux, const objnerf_render_multi_out* out) -> int {
    const int Sp = is_fine ? S + I : S;
    const float *hz[kMultiMaxSets], *hs[kMultiMaxSets], *hr[kMultiMaxSets];
    float* how[kMultiMaxSets];
    for (int k = 0; k < K; ++k) {
      float* z = is_fine ? w.zf(k) : w.zc(k);
      int rc;
      if (!is_fine) {
        // coarse depths are never perturbed here (multi_rendering.py:203-210)
        rc = objnerf_sample_coarse(in->h_rays[k], in->z_steps, nullptr, 0.f, cfg->use_disp, N, S, z, stream);
      } else {
        // importance sampling from the set's OWN weights of the joint compositing (multi_rendering.py:266-283)
        rc = objnerf_sample_pdf_merge_clip(w.zc(k), w.own(k), det ? in->u_det : in->u_rand + (int64_t)k * N * I, det ? 0 : I, N,
                                           S, I, 1e-5f, nullptr, z, in->h_clip ? in->h_clip[k] : nullptr, stream);
      }
      if (rc) return rc;
      // rays that missed their object's box (near = far = 0 => all depths 0): sigma is forced to -1e5 afterwards
      // (multi_rendering.py:40,83,92), i.e. exactly zero weight -- they are culled before the MLP kernel instead
      rc = objnerf_compact_rays(z, N, Sp, w.idx(k), w.count(K, k), w.scratch(K), stream);
      if (rc) return rc;
      const int oid = in->h_obj_ids[k];
      objnerf_mlp_args m;
      memset(&m, 0, sizeof(m));
      m.use_voxel = cfg->use_voxel;
      m.blob = blob; m.aux = aux;
      m.rays = in->h_rays[k]; m.z_vals = z; m.n_rays = N; m.S = Sp; m.grid = in->grid;
      m.ray_index = w.idx(k); m.n_active = w.count(K, k);
      if (oid > 0) {        // object branch with that id's code (multi_rendering.py:45-51, 63-69)
        m.do_object = 1; m.codes = in->code_table + (int64_t)oid * 64; m.code_stride = 0;
        m.inst_sigma = w.sigma(k); m.inst_rgb = w.rgb(k);
      } else {              // background: scene branch
        m.do_scene = 1; m.sigma = w.sigma(k); m.rgb = w.rgb(k);
      }
      if (w.hoist) {
        rc = objnerf_ray_bias(&m, w.ray_bias(k), stream);
        if (rc) return rc;
        m.ray_bias = w.ray_bias(k);
      }
      rc = objnerf_mlp_eval(&m, stream);
      if (rc) return rc;
      const bool use_boxes = oid == 0 && in->n_boxes > 0;                       // multi_rendering.py:239-241
      rc = objnerf_mask_sigma_rgb(w.sigma(k), w.rgb(k), in->h_rays[k], z, N, Sp, use_boxes ? in->boxes : nullptr,
                                  use_boxes ? in->n_boxes : 0, stream);
      if (rc) return rc;
      hz[k] = z; hs[k] = w.sigma(k); hr[k] = w.rgb(k); how[k] = w.own(k);
    }
    objnerf_composite_multi_args c;
    memset(&c, 0, sizeof(c));
    c.n_rays = N; c.K = K; c.S = Sp; c.h_z = hz; c.h_sigma = hs; c.h_rgb = hr;
    c.noise = is_fine ? in->noise_fine : in->noise_coarse; c.noise_std = cfg->noise_std; c.white_back = cfg->white_back;
    c.z_sorted = out->z_vals; c.weights = out->weights; c.obj_ids = out->obj_ids;
    c.opacity = out->opacity; c.rgb_map = out->rgb; c.depth = out->depth;
    c.h_own_weights = (!is_fine && I > 0) ? how : nullptr;
    c.scratch = objnerf_composite_multi_scratch_bytes(K, Sp) > 0 ? (void*)w.stage(K) : nullptr;
    return objnerf_composite_multi(&c, stream);
  };

  int rc = one_pass(false, in->blob_coarse, in->aux_coarse, coarse);
  if (rc || I <= 0) return rc;
  return one_pass(true, in->blob_fine, in->aux_fine, fine);
}

}  // extern "C"
