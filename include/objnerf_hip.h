/* objnerf_hip.h -- C ABI of libobjnerf_hip.so: the MI355X (gfx950) replacement for the
 * volume-rendering hot path of zju3dv/object_nerf.
 *
 * The reference has NO native interface for this path (it is pure PyTorch); each entry point
 * below names the reference Python code it replaces (paths relative to the reference root).
 * The binding a maintainer adds on the reference side is a ctypes stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless the name starts with `h_`;
 *   - `stream` is a hipStream_t passed as void*; calls only ENQUEUE work on it, never synchronise,
 *     never allocate: the caller owns every buffer including workspaces;
 *   - return 0 on success, negative on error; objnerf_last_error() gives the message of the last
 *     failing call made by the calling thread;
 *   - all arithmetic is fp32 with the reference's operation order wherever it is observable
 *     (DESIGN.md "numerics contract").
 */
#ifndef OBJNERF_HIP_H
#define OBJNERF_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OBJNERF_ABI_VERSION 10

int objnerf_abi_version(void);
const char* objnerf_last_error(void);

/* ---- sparse voxel grid: state of models/embedding_helper.py::EmbeddingVoxel (77-200) ---- */
typedef struct {
  const int32_t* idx_map;   /* (X,Y,Z) row-major; >=0 row of `table`, <0 empty (voxel_idx_map, 187-200) */
  const float* table;       /* (n_rows, 24) embedding_space_ftr.weight (81) */
  int32_t shape[3];         /* voxel_shape (111-122) */
  float offset[3];          /* voxel_offset = -bounds[0] (108) */
  float voxel_size;         /* voxel_size / scale_factor (102-103) */
  int32_t n_rows;
} objnerf_voxel_grid;

/* ---- packed MLP weights of one models/nerf_model.py::ObjectNeRF ---- */
/* sizes (in floats) of the weight stream and the aux block (biases + heads, then the hoisted weight columns of
 * objnerf_ray_bias as a compact matrix) */
int64_t objnerf_blob_floats(int use_voxel);
int64_t objnerf_aux_floats(void);
/* number of parameter tensors the packer consumes, and their canonical order:
 *   2*i = weight, 2*i+1 = bias of
 *   xyz_encoding_{1..8}.0, xyz_encoding_final, dir_encoding.0, sigma, rgb.0,
 *   instance_encoding_{1..4}.0, instance_encoding_final.0, inst_dir_encoding.0,
 *   instance_sigma, inst_rgb.0                      (names: nerf_model.py:41-58, 77-95) */
int objnerf_num_param_ptrs(void);
/* expected element count of parameter tensor `ptr_id` (for validation by the caller) */
int64_t objnerf_param_numel(int use_voxel, int ptr_id);
/* host: fill the two gather maps (uint32 per packed float) that describe the permutation
 * from the reference's (out,in) row-major nn.Linear tensors to the MFMA operand stream */
int objnerf_pack_index(int use_voxel, uint32_t* h_blob_idx, uint32_t* h_aux_idx);
/* device: blob[i] = param[idx>>24][idx & 0xFFFFFF] (0 where idx == 0xFFFFFFFF).
 * h_param_ptrs: HOST array of objnerf_num_param_ptrs() DEVICE pointers. */
int objnerf_pack_weights(int use_voxel, const uint32_t* blob_idx, const uint32_t* aux_idx,
                         const float* const* h_param_ptrs, float* blob, float* aux, void* stream);
/* ABI 10: the same for n_models (1 or 2) modules of one mode in ONE launch -- the coarse and fine model of a render_rays call
 * (train.py:45-51 builds both from one config).  h_param_ptrs: HOST array of n_models * objnerf_num_param_ptrs() DEVICE
 * pointers (model-major); h_blobs / h_auxs: HOST arrays of n_models DEVICE output pointers.  The product gathers on EVERY
 * call (SURVEY 8b "repack when params change": a cache keyed on host-visible parameter versions cannot see fused optimizers,
 * `.data` writes or graph-replayed steps; the launch costs ~8 us and is captured with the call in a graph).
 * objnerf_pack_weights is this with n_models = 1. */
int objnerf_pack_models(int use_voxel, const uint32_t* blob_idx, const uint32_t* aux_idx, int n_models,
                        const float* const* h_param_ptrs, float* const* h_blobs, float* const* h_auxs, void* stream);

/* Training only: the transposed weight stream of the hidden-to-hidden blocks, consumed by the fused backward of the
 * hidden chain (objnerf_train_args.blob_bwd).  Same tile/chunk format as the forward stream with rows = input
 * features and k = output features; the gradients w.r.t. the embeddings are not part of it. */
int64_t objnerf_bwd_blob_floats(void);
int objnerf_pack_index_bwd(int use_voxel, uint32_t* h_blob_idx);
/* objnerf_pack_index_bwd(mode, ...): 0 = plain-PE model, 1 = voxel model, 2 = voxel model + the embedding-gradient blocks
 * (objnerf_train_args.bwd_dx); objnerf_bwd_blob_floats() is the size of the longest of them. */
int objnerf_pack_weights_bwd(const uint32_t* blob_idx, const float* const* h_param_ptrs, float* blob, void* stream);
/* ---- stage entry points ---- */

/* coarse depths: models/rendering.py:260-277.  z_steps = torch.linspace(0,1,S) (S floats, device).
 * perturb_rand: (N,S) uniform [0,1) or NULL when perturb == 0. */
int objnerf_sample_coarse(const float* rays, const float* z_steps, const float* perturb_rand,
                          float perturb, int use_disp, int64_t n_rays, int S, float* z_vals,
                          void* stream);

/* Embedding.forward (embedding_helper.py:57-74): x (n, C) -> (n, C*(2F+1)) */
int objnerf_pos_encode(const float* x, int64_t n, int C, int n_freqs, float* out, void* stream);
/* the same with explicit frequency bands (n_freqs device floats): `logscale=False`, torch.linspace(1, 2^(F-1), F)
 * (embedding_helper.py:54-55); freqs == NULL is objnerf_pos_encode */
int objnerf_pos_encode_freqs(const float* x, int64_t n, int C, int n_freqs, const float* freqs, float* out, void* stream);

/* EmbeddingVoxel.forward (embedding_helper.py:325-411): xyz (n,3) -> scene_ftr (n,271), obj_ftr (n,104) */
int objnerf_voxel_embed(const objnerf_voxel_grid* grid, const float* xyz, int64_t n,
                        float* scene_ftr, float* obj_ftr, void* stream);

/* The fused encode + dual-branch MLP kernel.
 * Replaces the chunk loop models/rendering.py:86-137 (embedding_xyz -> ObjectNeRF.forward ->
 * ObjectNeRF.forward_instance) and render_tools/multi_rendering.py:30-93.
 *
 * Two input forms:
 *   fused (emb_xyz == NULL): points are rays_o + rays_d * z_vals[n, s] (rendering.py:279),
 *     embedded on the fly (voxel grid when use_voxel, else Embedding(3,10)); directions are
 *     embedded per ray; object codes are codes + ray * code_stride (code_stride 0 = one code).
 *   memory (emb_xyz != NULL): pre-embedded per-point inputs exactly as ObjectNeRF.forward /
 *     forward_instance receive them (nerf_model.py:97-152): emb_xyz (P,in_xyz), emb_dir (P,27),
 *     obj_voxel (P,104; voxel mode), obj_code (P,64).
 * Outputs (any may be NULL when its branch is off): sigma (P), rgb (P,3), inst_sigma (P),
 * inst_rgb (P,3); P = n_rays * S (fused) or n_points (memory). */
typedef struct {
  int32_t use_voxel;
  int32_t do_scene;          /* evaluate ObjectNeRF.forward */
  int32_t do_object;         /* evaluate ObjectNeRF.forward_instance */
  const float* blob;         /* objnerf_blob_floats() */
  const float* aux;          /* objnerf_aux_floats() */
  /* fused form */
  const float* rays;         /* (n_rays, 8) */
  const float* z_vals;       /* (n_rays, S) */
  int64_t n_rays;
  int32_t S;
  const float* codes;        /* embedding_instance */
  int64_t code_stride;       /* floats between consecutive rays' codes (64) or 0 */
  objnerf_voxel_grid grid;
  /* memory form */
  const float* emb_xyz;
  const float* emb_dir;
  const float* obj_voxel;
  const float* obj_code;
  int64_t n_points;
  /* outputs */
  float* sigma;
  float* rgb;
  float* inst_sigma;
  float* inst_rgb;
  /* evaluate just the density head of the selected branch (nerf_model.py:111-112, 142-143 `sigma_only=True`, used by
   * tools/extract_mesh.py:85-108): the final / direction / rgb layers are skipped (scene branch: 597,760 instead of
   * 699,904 MAC per point).  Memory form: as ObjectNeRF.forward gets its inputs; fused form: see `points` below */
  int32_t sigma_only;
  /* fused form only, optional (both or neither): evaluate a SUBSET of the rays.  ray_index: int32 ray numbers,
   * ascending or not; n_active: DEVICE pointer to how many of them are valid -- read by the kernel, so the count can
   * be produced on the stream (objnerf_compact_rays) without a host round trip.  Points of unlisted rays are neither
   * read nor written.  Replaces the "evaluate, then overwrite sigma" of rays that missed their object's box in
   * render_tools/multi_rendering.py:40,83,92. */
  const int32_t* ray_index;
  const int32_t* n_active;
  /* fused form only, optional: alpha compositing (models/rendering.py:139-229, eval mode: no occlusion mask, no noise)
   * begun in the kernel's epilogue -- a wave holds 32 consecutive samples of one ray, so it forms alpha, the
   * transmittance scan and the weighted sums of its 32-sample SEGMENT right where sigma / rgb were computed; they are
   * then never written (sigma / rgb / inst_sigma / inst_rgb may be NULL).  Requires do_scene, S % 32 == 0, no ray_index.
   *   comp_w    (n_rays, S) out: per-sample weights relative to the start of their segment -- of the scene set, or of
   *             the instance set when comp_inst_weights (rays_in_bbox, rendering.py:228-229); NULL = off
   *   comp_rec  (n_rays * S / 32, OBJNERF_SEG_REC_FLOATS) out: per segment [Q A R G B D - - | the same of the instance
   *             set]: product of (1 - alpha + 1e-10), sums of the local weights x {1, r, g, b, z}
   *   comp_last_delta  delta of the scene set's last sample: 1e10, or 0 with use_zero_as_last_delta (the instance set's is 0)
   * objnerf_composite_finish turns comp_w / comp_rec into the reference's weights and maps. */
  float* comp_w;
  float* comp_rec;
  float comp_last_delta;
  int32_t comp_inst_weights;
  /* fused form, optional: (n_rays, OBJNERF_RAY_BIAS_FLOATS) vectors written by objnerf_ray_bias for
   * the SAME blob / aux / rays / codes.  The parts of four layers' pre-activations that are constant along a ray -- the
   * object code's share of instance_encoding_1 / _3 and the direction embedding's share of dir_encoding /
   * inst_dir_encoding: the reference repeats both over the samples (rendering.py:89-94) -- are then taken from there
   * instead of being contracted per sample point: 2.45 % fewer MFMAs, same sums in another association. */
  const float* ray_bias;
  /* fused form with sigma_only (ABI 7): the density query of tools/extract_mesh.py:63-113 in ONE enqueue -- the kernel
   * takes the n_points sample POINTS themselves instead of rays x depths (rays / z_vals NULL, no direction, one branch:
   * do_scene xor do_object; an object query reads ONE code at `codes`), embeds them in registers (voxel grid or
   * Embedding(3,10), exactly as the render path does) and stops after the density head: sigma / inst_sigma (n_points).
   * The script's per-chunk embedding_xyz(...) -> forward(..., sigma_only=True) loop materialises 271 (+104) floats per
   * point and reads them back; here nothing but the 4-byte result reaches memory.  Either
   *   points   (n_points, 3) explicit positions, or
   *   lat_x / lat_y / lat_z  the axes of a lattice in np.stack(np.meshgrid(x, y, z), -1).reshape(-1, 3) order
   *            (extract_mesh.py:62-66; fp32 values as torch.FloatTensor(...) rounds them): point ((j * nx + i) * nz + k)
   *            = (x[i], y[j], z[k]), lat_n = {nx, ny, nz}, n_points = nx * ny * nz -- the 1.6 GB coordinate array of a
   *            512^3 grid is never built.
   * An object query may hoist its code like the render path does: `ray_bias` = ONE vector written by objnerf_ray_bias for
   * n_rays = 1 (any 8-float ray row, the query has no direction) with this `codes`: the code's share of
   * instance_encoding_1 / _3 is then a constant added in the layers' epilogues (64 of 439 / 567 inputs never contracted). */
  const float* points;
  const float* lat_x; const float* lat_y; const float* lat_z;
  int32_t lat_n[3];
  int32_t _pad_lat;
} objnerf_mlp_args;
#define OBJNERF_RAY_BIAS_FLOATS 448
/* out: objnerf_ray_bias_floats(n_rays) = n_rays * OBJNERF_RAY_BIAS_FLOATS floats -- the vectors for
 * objnerf_mlp_args.ray_bias.  Uses aux (the hoisted weight columns sit behind the aux block as a compact matrix, gathered by
 * objnerf_pack_weights), rays, codes / code_stride, n_rays, use_voxel, do_scene, do_object of `args`, and its ray subset
 * (ray_index / n_active) when given: only the listed rays' vectors are written. */
int64_t objnerf_ray_bias_floats(int64_t n_rays);
int objnerf_ray_bias(const objnerf_mlp_args* args, float* out, void* stream);
#define OBJNERF_SEG_REC_FLOATS 16
int objnerf_mlp_eval(const objnerf_mlp_args* args, void* stream);

/* scene + instance alpha compositing: models/rendering.py:139-229.
 * noise / noise_inst: (N,S) N(0,1) draws already multiplied by nothing (kernel multiplies by
 * noise_std) or NULL when noise_std == 0.  pass_through_mask: (N) uint8 or NULL. */
typedef struct {
  int64_t n_rays;
  int32_t S;
  const float* z_vals;       /* (N,S) */
  const float* sigma;        /* (N,S) */
  const float* rgb;          /* (N,S,3) */
  const float* inst_sigma;   /* NULL when forward_instance is off */
  const float* inst_rgb;
  const float* noise;
  const float* noise_inst;
  float noise_std;
  int32_t white_back;
  int32_t use_zero_as_last_delta;
  int32_t occlusion;         /* (not is_eval) and frustum_bound_th > 0 */
  float frustum_bound_th;
  const uint8_t* pass_through_mask;
  int32_t rays_in_bbox;      /* weights_out := instance weights (rendering.py:228-229) */
  /* outputs */
  float* weights;            /* (N,S) */
  float* opacity;            /* (N) */
  float* rgb_map;            /* (N,3) */
  float* depth;              /* (N) */
  float* rgb_inst;           /* (N,3) */
  float* depth_inst;         /* (N) */
  float* opacity_inst;       /* (N) */
} objnerf_composite_args;
int objnerf_composite(const objnerf_composite_args* args, void* stream);
/* Second half of the fused form (objnerf_mlp_args.comp_*): per ray, the segments' incoming transmittances in ascending
 * order, weights[n, s] *= T(segment of s) in place (weights = the comp_w the MLP kernel wrote), and the maps of
 * rendering.py:164-229 (rgb_map white-backed when white_back; the instance maps, always white-backed, when has_instance;
 * inst_weights: `weights` are the instance set's, objnerf_mlp_args.comp_inst_weights).
 * Results are BIT-EQUAL to objnerf_composite on the same sigma / rgb (shared arithmetic, csrc/composite_seg.h).
 * HBM traffic: 8 B per sample + 64 B per 32 samples, against 40 B per sample for objnerf_composite + 32 B per sample of
 * sigma / rgb stores in the MLP kernel. */
int objnerf_composite_finish(const float* seg_records, int64_t n_rays, int S, int has_instance, int inst_weights,
                             int white_back, float* weights, float* opacity, float* rgb_map, float* depth, float* rgb_inst,
                             float* depth_inst, float* opacity_inst, void* stream);

/* sample_pdf + sort(cat) : models/rendering.py:11-61 and 302-313.
 * weights: (N,S) coarse weights (the kernel uses weights[:,1:-1] and z_mid of z_coarse);
 * u: (I) = torch.linspace(0,1,I) when det (u_stride 0) or (N,I) uniform draws (u_stride I).
 * z_samples (N,I) optional output of sample_pdf alone; z_fine (N,S+I) ascending. */
int objnerf_sample_pdf_merge(const float* z_coarse, const float* weights, const float* u,
                             int64_t u_stride, int64_t n_rays, int S, int I, float eps,
                             float* z_samples, float* z_fine, void* stream);

/* the same with the depth clip of render_tools/multi_rendering.py:277-285 (ray sets with 10 columns): clip (N,2) =
 * (bbox_mask_near, bbox_mask_far) per ray; merged depths strictly inside that interval are moved to its upper end.
 * clip == NULL: identical to objnerf_sample_pdf_merge. */
int objnerf_sample_pdf_merge_clip(const float* z_coarse, const float* weights, const float* u,
                                  int64_t u_stride, int64_t n_rays, int S, int I, float eps,
                                  float* z_samples, float* z_fine, const float* clip, void* stream);

/* standalone sample_pdf (rendering.py:11-61): bins (N,nb), weights (N,nb-1) -> (N,I) */
int objnerf_sample_pdf(const float* bins, const float* weights, const float* u, int64_t u_stride,
                       int64_t n_rays, int nb, int I, float eps, float* samples, void* stream);

/* ---- multi-object path: render_tools/multi_rendering.py ---- */

/* Oriented boxes (utils/bbox_utils.py::BBoxRayHelper): OBJNERF_BOX_DOUBLES float64 per box, device array:
 *   [0] scale_factor, [1..9] pose_avg R (row-major), [10..12] pose_avg t, [13..21] axis_align_mat R,
 *   [22..24] axis_align_mat t, [25..27] bbox min, [28..30] bbox max (after the caller applied
 *   bbox_enlarge, bbox_utils.py:171-181).  The transform runs in float64 like the numpy code it
 *   replaces (bbox_utils.py:119-130), the comparison in fp32 like the torch code (169-186). */
#define OBJNERF_BOX_DOUBLES 31
/* sigma[n, s] = -1e5 where z_vals[n, S-1] == 0 (multi_rendering.py:40,83,92) and, when n_boxes > 0,
 * where the sample point rays_o + rays_d * z lies inside any box (multi_rendering.py:239-241). */
int objnerf_mask_sigma(float* sigma, const float* rays, const float* z_vals, int64_t n_rays, int S,
                       const double* boxes, int n_boxes, void* stream);
/* the same, and additionally rgb[n, s, :] = 0 on the rays with z_vals[n, S-1] == 0 (rgb may be NULL): for rays that
 * objnerf_compact_rays culled before the MLP kernel, whose sigma / rgb were never written */
int objnerf_mask_sigma_rgb(float* sigma, float* rgb, const float* rays, const float* z_vals, int64_t n_rays, int S,
                           const double* boxes, int n_boxes, void* stream);
/* Ray culling without a host round trip: ray_index[0 .. *n_active) = the rays with z_vals[n, S-1] != 0 (the complement of
 * multi_rendering.py:40's zero_mask), ascending; both written on the stream.  scratch: objnerf_compact_scratch_ints(n_rays)
 * int32.  Feed ray_index / n_active to objnerf_mlp_args. */
int64_t objnerf_compact_scratch_ints(int64_t n_rays);
int objnerf_compact_rays(const float* z_vals, int64_t n_rays, int S, int32_t* ray_index, int32_t* n_active,
                         int32_t* scratch, void* stream);
/* check_in_any_boxes (bbox_utils.py:189-207) on explicit points: xyz (n,3) -> inside (n) uint8 */
int objnerf_points_in_boxes(const float* xyz, int64_t n, const double* boxes, int n_boxes,
                            uint8_t* inside, void* stream);

/* volume_rendering_multi (multi_rendering.py:96-157): K ray sets of S samples each, given as
 * arrays of K device pointers (host arrays).  Joint stable ascending sort by z, last delta 0.
 * Outputs: z_sorted (N,K*S), weights (N,K*S), obj_ids (N,K*S) float (index into the K sets),
 * opacity (N), rgb_map (N,3), depth (N); own_weights: optional K host pointers to (N,S) buffers
 * receiving each set's weights in its own sample order (multi_rendering.py:269-271). */
typedef struct {
  int64_t n_rays;
  int32_t K;
  int32_t S;
  const float* const* h_z;      /* K x (N,S) */
  const float* const* h_sigma;  /* K x (N,S) */
  const float* const* h_rgb;    /* K x (N,S,3) */
  const float* noise;           /* (N,K*S) or NULL */
  float noise_std;
  int32_t white_back;
  float* z_sorted;
  float* weights;
  float* obj_ids;               /* may be NULL */
  float* opacity;
  float* rgb_map;
  float* depth;
  float* const* h_own_weights;  /* may be NULL */
  /* K*S samples of 28 bytes are staged per pixel: in LDS up to 152 KiB (K*S <= 5,558), beyond that in this device
   * buffer of objnerf_composite_multi_scratch_bytes(K, S) bytes (0 when LDS suffices; may then be NULL).  The reference
   * sorts any K*S (multi_rendering.py:112).  1 <= K <= 64. */
  void* scratch;
} objnerf_composite_multi_args;
int64_t objnerf_composite_multi_scratch_bytes(int K, int S);
int objnerf_composite_multi(const objnerf_composite_multi_args* args, void* stream);

/* Ray generation for the editor: one kernel replaces get_ray_directions + get_rays
 * (datasets/ray_utils.py:5-51), EditableRenderer.generate_rays (render_tools/editable_renderer.py:153-181) and
 * the CPU numba slab test behind it (utils/bbox_utils.py:100-117,132-156; datasets/geo_utils.py:111-162).
 * h_c2w: HOST 12 floats, the (3,4) camera-to-object matrix `Toc` (translation already / scale_factor).
 * h_box: HOST OBJNERF_BOX_DOUBLES doubles or NULL.  NULL: near/far are the given constants (background set);
 * else near/far = ray/box entry/exit in float64 (bounds grown by bbox_enlarge when > 0) / scale_factor, and 0/0
 * for rays that miss the box or start inside it.  rays: (H*W, 8) row-major pixel order. */
int objnerf_generate_rays(int H, int W, float focal, const float* h_c2w, float near, float far,
                          const double* h_box, double bbox_enlarge, float* rays, void* stream);
/* The same for a subset of the image rows (one rank's share of a sharded frame): n_rows rows, local row lr = image row
 * row0 + (lr / row_block) * row_block * block_stride + lr % row_block.  A contiguous band is row_block = n_rows,
 * block_stride = 1; the block-cyclic share of rank r of w ranks is row0 = r * row_block, block_stride = w.
 * rays: (n_rows * W, 8). */
int objnerf_generate_rays_rows(int H, int W, float focal, const float* h_c2w, float near, float far,
                               const double* h_box, double bbox_enlarge, int row0, int n_rows, int row_block,
                               int block_stride, float* rays, void* stream);
/* The stage-by-stage form of the above, as the reference's unchanged caller issues it
 * (render_tools/editable_renderer.py:191-198, 215, 257, 163-170) -- same device arithmetic as objnerf_generate_rays:
 *   objnerf_ray_directions    datasets/ray_utils.py:5-25   get_ray_directions -> directions (H*W, 3)
 *   objnerf_get_rays          datasets/ray_utils.py:28-51  get_rays: directions (n,3), c2w = DEVICE pointer to the (3,4)
 *                             matrix, rows row_stride floats apart (4 for a contiguous (3,4) or the top of a (4,4))
 *                             -> rays_o (n,3), rays_d (n,3)
 *   objnerf_ray_box_near_far  utils/bbox_utils.py:132-156 BBoxRayHelper.get_ray_bbox_intersections (+ the slab test
 *                             datasets/geo_utils.py:111-162 it calls on the host): rays_o/rays_d (n,3), h_box = HOST
 *                             OBJNERF_BOX_DOUBLES doubles (bounds NOT enlarged), bbox_enlarge grows both bounds when > 0
 *                             -> hit (n) uint8, near (n), far (n) already divided by the box's scale_factor, 0/0 on a miss */
int objnerf_ray_directions(int H, int W, float focal, float* directions, void* stream);
int objnerf_get_rays(const float* directions, int64_t n, const float* c2w, int row_stride, float* rays_o, float* rays_d,
                     void* stream);
int objnerf_ray_box_near_far(const float* rays_o, const float* rays_d, int64_t n, const double* h_box, double bbox_enlarge,
                             uint8_t* hit, float* near, float* far, void* stream);

/* ---- whole render_rays (models/rendering.py:233-337) in one enqueue ---- */
typedef struct {
  int32_t use_voxel;
  int32_t N_samples;
  int32_t N_importance;
  int32_t use_disp;
  float perturb;
  float noise_std;
  int32_t white_back;
  int32_t forward_instance;
  int32_t is_eval;
  int32_t use_zero_as_last_delta;
  float frustum_bound_th;
  int32_t rays_in_bbox;
  /* 0 (default): a pass without occlusion mask and noise whose sample count is a multiple of 32 composites in the MLP
   * kernel's epilogue (objnerf_mlp_args.comp_*: sigma / rgb never reach memory, workspace 2 B instead of 32 B per sample);
   * 1: always the two-kernel form (MLP kernel -> sigma / rgb in the workspace -> objnerf_composite).  Results are
   * bit-equal either way. */
  int32_t separate_composite;
  /* 0 (default): the passes take the per-ray constant terms from objnerf_ray_bias (objnerf_mlp_args.ray_bias;
   * 1792 B of workspace per ray); 1: every term contracted per sample point as in round 2 (A/B switch) */
  int32_t no_hoist;
} objnerf_render_cfg;

typedef struct {
  /* (N,S_typ) */ float* weights; float* z_vals;
  /* (N) */ float* opacity; float* depth; float* depth_instance; float* opacity_instance;
  /* (N,3) */ float* rgb; float* rgb_instance;
} objnerf_render_out;

typedef struct {
  const float* rays;               /* (N,8) */
  int64_t n_rays;
  const float* codes;              /* embedding_instance (N,64) (or one code with stride 0) */
  int64_t code_stride;
  const uint8_t* pass_through_mask;/* (N) or NULL */
  const float* blob_coarse; const float* aux_coarse;
  const float* blob_fine; const float* aux_fine;      /* NULL when N_importance == 0 */
  objnerf_voxel_grid grid;
  const float* z_steps;            /* linspace(0,1,N_samples) */
  const float* u_det;              /* linspace(0,1,N_importance) (det) */
  /* random inputs, only read when perturb > 0 / noise_std > 0 (caller draws them):
   * perturb_rand (N,S), u_rand (N,I), noise[4] = coarse scene, coarse inst, fine scene, fine inst */
  const float* perturb_rand; const float* u_rand; const float* noise[4];
  void* workspace;                 /* objnerf_render_workspace_bytes() */
} objnerf_render_in;

int64_t objnerf_render_workspace_bytes(const objnerf_render_cfg* cfg, int64_t n_rays);
int objnerf_render_rays(const objnerf_render_cfg* cfg, const objnerf_render_in* in,
                        const objnerf_render_out* coarse, const objnerf_render_out* fine,
                        void* stream);

/* ---- whole render_rays_multi (render_tools/multi_rendering.py:160-325) in one enqueue ----
 * K ray sets of the same pixels (obj id 0 = background: scene branch; id > 0: object branch with row `id` of the code
 * table), per set coarse depths -> culling of the rays that missed their box (objnerf_compact_rays; no host round
 * trip) -> one branch of the fused MLP kernel over the surviving rays -> sigma masks (culled rays, samples inside the
 * removed objects' boxes for id 0) -> joint depth-sorted compositing -> per-set importance sampling from the set's own
 * weights -> the same again with the fine model.  Nothing is synchronised or allocated. */
typedef struct {
  int32_t use_voxel;
  int32_t N_samples;
  int32_t N_importance;
  int32_t use_disp;
  float perturb;             /* != 0: the importance samples use u_rand instead of linspace (multi_rendering.py:276) */
  float noise_std;
  int32_t white_back;
  int32_t no_hoist;          /* as objnerf_render_cfg.no_hoist */
} objnerf_render_multi_cfg;

typedef struct {
  int64_t n_rays;
  int32_t K;
  const float* const* h_rays;      /* HOST array of K device pointers, each (N,8) [o, d, near, far] */
  const int32_t* h_obj_ids;        /* HOST array of K ids */
  const float* code_table;         /* (N_max_objs, 64) code_library.embedding_instance.weight (multi_rendering.py:46) */
  const float* blob_coarse; const float* aux_coarse;
  const float* blob_fine; const float* aux_fine;
  objnerf_voxel_grid grid;
  const float* z_steps;            /* linspace(0,1,N_samples) */
  const float* u_det;              /* linspace(0,1,N_importance) */
  const float* u_rand;             /* (K,N,I) uniform draws, read when perturb != 0 */
  const float* noise_coarse;       /* (N,K*S) N(0,1) draws, read when noise_std != 0 */
  const float* noise_fine;         /* (N,K*(S+I)) */
  const float* const* h_clip;      /* NULL, or HOST array of K device pointers, each NULL or (N,2): columns 8:10 of a
                                    * 10-column ray set (multi_rendering.py:277-285, objnerf_sample_pdf_merge_clip) */
  const double* boxes;             /* n_boxes x OBJNERF_BOX_DOUBLES: background_skip_bbox, applied to id-0 sets */
  int32_t n_boxes;
  void* workspace;                 /* objnerf_render_multi_workspace_bytes() */
} objnerf_render_multi_in;

typedef struct {
  /* (N, K*S_typ) */ float* z_vals; float* weights; float* obj_ids;   /* obj_ids: coarse pass only, may be NULL */
  /* (N) */ float* opacity; float* depth;
  /* (N,3) */ float* rgb;
} objnerf_render_multi_out;

int64_t objnerf_render_multi_workspace_bytes(const objnerf_render_multi_cfg* cfg, int32_t K, int64_t n_rays);
int objnerf_render_rays_multi(const objnerf_render_multi_cfg* cfg, const objnerf_render_multi_in* in,
                              const objnerf_render_multi_out* coarse, const objnerf_render_multi_out* fine,
                              void* stream);

/* ---- training path: differentiable stages behind train.py:147-180 ---- */

/* C[M,N] = A'[M,K] * B'[K,N] (+ C) with optional bias / LeakyReLU / sigmoid epilogue: the fp32 MFMA GEMM of the
 * layer-wise training path (csrc/gemm.h).  A'[m][k] = a_k_contig ? A[m*lda+k] : A[k*lda+m];
 * B'[k][n] = b_k_contig ? B[n*ldb+k] : B[k*ldb+n].  split_k > 1 adds with fp32 atomics (C must hold the
 * initial value).  epilogue: 0 none, 1 +bias, 2 +bias+LeakyReLU(0.01), 3 +bias+sigmoid, 4 LeakyReLU backward:
 * `bias` then points at the layer's saved output Y[M,N] (leading dimension ldc) and C = Y > 0 ? C : 0.01 C. */
int objnerf_gemm(const float* A, int64_t lda, int a_k_contig, const float* B, int64_t ldb, int b_k_contig,
                 float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int accumulate, int epilogue,
                 const float* bias, int split_k, void* stream);

/* ObjectNeRF.forward + forward_instance (nerf_model.py:97-152) on pre-embedded dense inputs, keeping the
 * activations for the backward pass.  h_params: HOST array of objnerf_num_param_ptrs() DEVICE pointers to the
 * raw nn.Linear tensors (same order as objnerf_pack_weights; nothing is packed on this path). */
typedef struct {
  int32_t use_voxel;
  int32_t do_object;
  int64_t n_points;
  const float* const* h_params;
  const float* emb_xyz;     /* (P, in_xyz) */
  const float* emb_dir;     /* (P, 27) */
  const float* obj_voxel;   /* (P, 104) voxel mode */
  const float* obj_code;    /* (P, 64) */
  float* sigma;             /* (P) */
  float* rgb;               /* (P,3) */
  float* inst_sigma;        /* (P) */
  float* inst_rgb;          /* (P,3) */
  float* workspace;         /* objnerf_train_workspace_floats(): saved activations, input of the backward */
  /* optional (both or neither): the packed weight stream of the SAME parameter values (objnerf_pack_weights).  When
   * given, the forward runs on the persistent MFMA kernel of objnerf_mlp_eval (memory form), which additionally
   * writes every layer's activations -- one launch per branch instead of one GEMM per layer.  The backward is
   * the same either way.
   * CONTRACT with the backward: a forward that ran WITH blob also leaves the LeakyReLU sign masks behind the activation matrices
   * of `workspace`, and a backward called with blob != NULL reads them there (its dgrad chain is fed by the masks).  Pass the
   * SAME blob / no-blob choice to both calls of a pair: a backward given a blob after a layer-by-layer forward (blob == NULL)
   * would read masks nobody wrote. */
  const float* blob; const float* aux;
  /* optional, read by objnerf_mlp_train_backward only (needs aux too): objnerf_pack_weights_bwd() of the same
   * parameter values.  When given, the dgrad chain through the hidden layers runs in one persistent MFMA kernel
   * (gradient tiles stay in registers from layer to layer) instead of one GEMM + activation-backward per layer. */
  const float* blob_bwd;
  /* optional, forward only, with blob/aux: the un-embedded inputs of the same points (points = rays x S in ray-major
   * order, exactly what emb_xyz / emb_dir / obj_voxel / obj_code were computed from).  When rays != NULL the forward
   * computes the embeddings in registers like objnerf_mlp_eval's fused form instead of reading them back. */
  const float* rays; const float* z_vals; int64_t n_rays; int32_t S;
  /* (ABI 10; was padding) read by objnerf_mlp_train_backward only: blob_bwd was packed through objnerf_pack_index_bwd MODE 2, i.e.
   * it also carries the embedding-column blocks of xyz_encoding_1 / _5 and instance_encoding_1 / _3, and the fused chain forms
   * d_emb_xyz / d_obj_voxel itself while those layers' gradient tiles are in registers (needs the forward's masks: blob given to
   * the forward too).  0: the mode-1 stream, the embedding gradients are two segmented GEMMs after the chain (rounds 2-5). */
  int32_t bwd_dx;
  const float* codes; int64_t code_stride;
  objnerf_voxel_grid grid;
  /* optional (ABI 8), read by objnerf_mlp_train_backward only, voxel mode: the sample positions (P,3) the embeddings were
   * taken at and the feature table's gradient (n_rows, 24; accumulated into).  When both are given the call also does
   * objnerf_voxel_embed_backward's work for this pass -- the scatter of d_emb_xyz / d_obj_voxel into the table -- right
   * after the embedding-gradient products (one call less per pass; on a side stream beside the weight-gradient kernels it
   * measured slower, CHANGELOG.md). */
  const float* scatter_xyz; float* scatter_table_grad;
  /* optional (ABI 9), read by objnerf_mlp_train_backward only: the direction embedding per RAY (n_rays, 27).  Together with
   * n_rays / S (S % 16 == 0, n_rays * S == n_points) and -- when do_object -- codes / code_stride (= 64) it selects the per-ray
   * form of the terms that are constant along a ray (the reference repeats rays_d and the code over the samples,
   * models/rendering.py:89-94): the gradients of the weight columns that meet the direction embedding / the object code, and the
   * gradient w.r.t. the code, are contracted over n_points / 16 segment sums instead of n_points points.  emb_dir / obj_code
   * (per point) may then be NULL, and d_obj_code receives (n_points / 16, 64): one row per 16 consecutive points, to be summed
   * over a ray's S / 16 segments by the caller (objnerf_sum_over_samples with S / 16).  The CALLER selects the form by passing
   * emb_dir_ray or NULL (round 6: the library no longer consults OBJNERF_TRAIN_PER_RAY / OBJNERF_WGRAD for it). */
  const float* emb_dir_ray;
  /* optional (ABI 9), forward only, with rays / blob / aux: n_rays * OBJNERF_RAY_BIAS_FLOATS floats of scratch.  When given, the
   * forward takes the per-ray constant terms from objnerf_ray_bias (written there by this call) and skips their k-steps, as
   * the inference passes do (objnerf_mlp_args.ray_bias): 2.45 % fewer MFMAs, sums in another association. */
  float* ray_bias_ws;
} objnerf_train_args;
int64_t objnerf_train_workspace_floats(int do_object, int64_t n_points);
/* scratch of objnerf_mlp_train_backward: the gradients w.r.t. every layer's pre-activation output (12.9 KB per point) + the
 * partial tiles of the grouped weight-gradient pass (64 tiles x ~P/1640 slices x 66 KB: 1.0 GB at the reference batch of
 * 393,216 points) + the head kernels' partial sums */
int64_t objnerf_train_scratch_floats(int64_t n_points);
int objnerf_mlp_train_forward(const objnerf_train_args* args, void* stream);
/* Backward of the call above (same args, outputs and workspace untouched in between).
 * d_*: gradients w.r.t. sigma (P), rgb (P,3), inst_sigma, inst_rgb.  h_param_grads: HOST array of DEVICE
 * pointers, one per parameter tensor, ACCUMULATED into (+=).  d_emb_xyz (P,in_xyz; only the 208 voxel-feature columns
 * are written -- the xyz positional-encoding columns have no consumer, depths are detached -- and nothing in plain-PE
 * mode), d_obj_voxel (P,104),
 * d_obj_code (P,64; (P/16,64) in the per-ray form, see emb_dir_ray) are overwritten.  scratch: objnerf_train_scratch_floats() floats (holds the gradient w.r.t. every
 * layer's pre-activation output, in the layout of the activation workspace). */
int objnerf_mlp_train_backward(const objnerf_train_args* args, const float* d_sigma, const float* d_rgb,
                               const float* d_inst_sigma, const float* d_inst_rgb, float* const* h_param_grads,
                               float* d_emb_xyz, float* d_obj_voxel, float* d_obj_code, float* scratch,
                               void* stream);

/* Backward of objnerf_composite (models/rendering.py:139-229): same `fwd` arguments as the forward call (its
 * outputs are not read; weights/alphas are recomputed).  g_*: gradients of the maps (any may be NULL = zero):
 * rgb_map (N,3), depth (N), opacity (N), rgb_inst (N,3), depth_inst (N), opacity_inst (N).  Outputs (overwritten):
 * d_sigma (N,S), d_rgb (N,S,3), d_inst_sigma (N,S), d_inst_rgb (N,S,3) (instance ones only with instance inputs).
 * No gradient flows to z_vals or through the occlusion mask (a comparison), as in the reference. */
int objnerf_composite_backward(const objnerf_composite_args* fwd, const float* g_rgb_map, const float* g_depth,
                               const float* g_opacity, const float* g_rgb_inst, const float* g_depth_inst,
                               const float* g_opacity_inst, float* d_sigma, float* d_rgb, float* d_inst_sigma,
                               float* d_inst_rgb, void* stream);

/* Backward of objnerf_voxel_embed w.r.t. the feature table (embedding_space_ftr.weight): table_grad
 * (n_rows,24) += scatter of d_scene_ftr (n,271) / d_obj_ftr (n,104) through the positional-encoding
 * derivative and the trilinear weights (fp32 atomics).  No gradient w.r.t. xyz (none is needed: depths are
 * detached, rendering.py:307). */
int objnerf_voxel_embed_backward(const objnerf_voxel_grid* grid, const float* xyz, int64_t n,
                                 const float* d_scene_ftr, const float* d_obj_ftr, float* table_grad, void* stream);

/* out[r, c] += sum_{s < S} x[r*S + s, c]   (gradient of the `repeat` of per-ray codes, rendering.py:94) */
int objnerf_sum_over_samples(const float* x, int64_t n_rays, int S, int C, float* out, void* stream);
/* Backward of the row gather CodeLibrary.forward does (models/code_library.py:20-28: nn.Embedding on `instance_ids`):
* table_grad (n_table_rows, C) += the rows d_rows (n, C) that picked each table row, added in a fixed order (blocks of 128 ids dealt
 * to 16 waves, ascending inside a wave, the waves' sums in wave order: bit-reproducible, no atomics).  ids: int64 (n), values outside [0, n_table_rows) are ignored.  C <= 1024. */
int objnerf_rows_gather_backward(const float* d_rows, const int64_t* ids, int64_t n, int C, int64_t n_table_rows, float* table_grad,
                                 void* stream);

/* xyz[n, s, :] = rays_o + rays_d * z_vals[n, s]  (rendering.py:279), materialised for the training path */
int objnerf_sample_points(const float* rays, const float* z_vals, int64_t n_rays, int S, float* xyz, void* stream);

/* ---- any architecture `config.model` can describe (models/nerf_model.py:18-95) ----
 * objnerf_mlp_eval's persistent kernel is specialised for the architecture of every shipped reference config (D 8, W 256,
 * skips [4], inst_D 4, inst_W 128, inst_skips [2], 10 / 4 / 6 frequencies, 16 + 8 voxel channels, 64-d code).  Other shapes run
 * layer by layer on the fp32 MFMA GEMM (bias / LeakyReLU / sigmoid epilogues, torch.cat inputs as column blocks), on the
 * reference's own nn.Linear tensors (no packing), activations in a caller-provided workspace -- still no CPU / PyTorch path. */
typedef struct {
  int32_t D, W;                    /* scene branch: xyz_encoding_1..D (nerf_model.py:41-51) */
  int32_t n_skips; int32_t skips[8];          /* layer indices i (0-based, as config.model.skips) whose input is cat([input, h]) */
  int32_t inst_D, inst_W;          /* object branch: instance_encoding_1..inst_D (77-88) */
  int32_t n_inst_skips; int32_t inst_skips[8];
  int32_t in_xyz;                  /* columns of emb_xyz  (in_channels_xyz, 25-35) */
  int32_t in_dir;                  /* columns of emb_dir  (in_channels_dir) */
  int32_t obj_voxel_c;             /* columns of obj_voxel (0 without voxel embedding) */
  int32_t code_c;                  /* columns of obj_code (N_obj_code_length) */
} objnerf_arch;
/* parameter table: HOST array of objnerf_arch_num_param_ptrs() DEVICE pointers, (weight, bias) pairs of
 *   xyz_encoding_{1..D}.0, xyz_encoding_final, dir_encoding.0, sigma, rgb.0,
 *   instance_encoding_{1..inst_D}.0, instance_encoding_final.0, inst_dir_encoding.0, instance_sigma, inst_rgb.0
 * in nn.Linear (out, in) row-major layout */
int objnerf_arch_num_param_ptrs(const objnerf_arch* arch);
typedef struct {
  objnerf_arch arch;
  const float* const* h_params;
  int32_t do_scene, do_object, sigma_only, _pad;
  int64_t n_points;
  const float* emb_xyz;            /* (P, in_xyz) */
  const float* emb_dir;            /* (P, in_dir); may be NULL with sigma_only */
  const float* obj_voxel;          /* (P, obj_voxel_c) */
  const float* obj_code;           /* (P, code_c) */
  float* sigma; float* rgb; float* inst_sigma; float* inst_rgb;       /* (P), (P,3), (P), (P,3) */
  float* workspace;                /* objnerf_mlp_generic_workspace_floats() */
} objnerf_mlp_generic_args;
int64_t objnerf_mlp_generic_workspace_floats(const objnerf_arch* arch, int64_t n_points);
int objnerf_mlp_generic(const objnerf_mlp_generic_args* args, void* stream);
/* Training of such architectures: objnerf_mlp_generic with every layer's output kept (workspace: activations, D*W + W + W/2
 * (+ inst_D*inst_W + inst_W + inst_W/2) floats per point), and its backward -- dgrad GEMMs with the LeakyReLU backward in the
 * epilogue, split-K weight / bias gradients ACCUMULATED (+=, fp32 atomics) into h_param_grads (same order as h_params), and the
 * gradients w.r.t. the inputs: the first emb_cols columns of emb_xyz (P, emb_cols) (the voxel-feature part; 0 = none), obj_voxel
 * (P, obj_voxel_c), obj_code (P, code_c), all overwritten.  d_*: gradients w.r.t. sigma (P), rgb (P,3), inst_sigma, inst_rgb.
 * The default architecture trains on the fused kernels (objnerf_mlp_train_forward / _backward). */
int64_t objnerf_mlp_generic_train_workspace_floats(const objnerf_arch* arch, int64_t n_points);
int64_t objnerf_mlp_generic_train_scratch_floats(const objnerf_arch* arch, int64_t n_points);
int objnerf_mlp_generic_train_forward(const objnerf_mlp_generic_args* args, void* stream);
int objnerf_mlp_generic_train_backward(const objnerf_mlp_generic_args* args, const float* d_sigma, const float* d_rgb,
                                       const float* d_inst_sigma, const float* d_inst_rgb, float* const* h_param_grads,
                                       float* d_emb_xyz, int emb_cols, float* d_obj_voxel, float* d_obj_code, float* scratch,
                                       void* stream);
/* backward of objnerf_pos_encode_block w.r.t. x: d_x (n, C; row stride ldd) = d_out[:, :C] + sum_k f_k (cos(f_k x) d_sin_k -
 * sin(f_k x) d_cos_k); and of objnerf_voxel_features w.r.t. the table: table_grad (n_rows, C) += trilinear scatter of d_raw */
int objnerf_pos_encode_block_backward(const float* x, int64_t ldx, int64_t n, int C, int n_freqs, const float* freqs,
                                      const float* d_out, int64_t ldo, float* d_x, int64_t ldd, void* stream);
int objnerf_voxel_features_backward(const objnerf_voxel_grid* grid, int C, const float* xyz, int64_t n, const float* d_raw,
                                    int64_t ldd, float* table_grad, void* stream);
/* building blocks of the embeddings of such architectures:
 *   objnerf_voxel_features   the trilinear sparse-voxel lookup of EmbeddingVoxel (embedding_helper.py:331-389) BEFORE the
 *                            positional encoding, for a table of C channels per row: xyz (n,3) -> out (n, C), row stride ldo
 *   objnerf_pos_encode_block Embedding.forward on a column block: x (n, C) with row stride ldx -> out (n, C*(2F+1)) with row
 *                            stride ldo (each part of a torch.cat([...], -1) is written into its own columns); freqs NULL =
 *                            bands 2^k, else n_freqs device floats (logscale=False)
 *   objnerf_repeat_rows      out[r] = src[r / repeat]: a per-ray row (direction embedding, object code) repeated over the
 *                            ray's samples (rendering.py:89-94) */
int objnerf_voxel_features(const objnerf_voxel_grid* grid, int C, const float* xyz, int64_t n, float* out, int64_t ldo, void* stream);
int objnerf_pos_encode_block(const float* x, int64_t ldx, int64_t n, int C, int n_freqs, const float* freqs, float* out,
                             int64_t ldo, void* stream);
int objnerf_repeat_rows(const float* src, int64_t lds, int64_t n_rows, int C, int repeat, float* out, int64_t ldo, void* stream);

/* ---- measurement hooks (bench.py): HIP-event timing of the MLP kernel on `stream` ---- */
/* When enabled, objnerf_mlp_eval brackets its launch with hipEvents; objnerf_timing_read
 * synchronises those events and returns {launch count, total ms} since the last reset. */
int objnerf_timing_enable(int on);
int objnerf_timing_read(int64_t* launches, double* total_ms);
/* The same for the training calls: when enabled, objnerf_mlp_train_forward / _backward bracket their phases with hipEvents
 * on `stream`; objnerf_train_timing_read synchronises them and returns, per phase, the summed milliseconds and the number
 * of spans since the last reset (both arrays OBJNERF_TRAIN_PHASES long).  Phases: 0 forward (fused MLP forward that keeps the
 * activations), 1 dgrad (the chain through the hidden layers), 2 dX (gradients w.r.t. the embeddings), 3 voxel-table scatter,
 * 4 weight gradients. */
#define OBJNERF_TRAIN_PHASES 5
int objnerf_train_timing_enable(int on);
int objnerf_train_timing_read(double* ms_by_phase, int64_t* spans_by_phase);

#ifdef __cplusplus
}
#endif
#endif /* OBJNERF_HIP_H */
