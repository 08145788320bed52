"""ctypes binding of libobjnerf_hip.so (include/objnerf_hip.h).

The library is the product: there is NO CPU / PyTorch fallback.  If the shared object is
missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import functools
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# OBJNERF_LIB: developer hook for A/B-timing build variants (tools/); the product library is the in-tree one
LIB_PATH = os.environ.get("OBJNERF_LIB") or os.path.join(_HERE, "libobjnerf_hip.so")
ABI_VERSION = 10    # OBJNERF_ABI_VERSION of include/objnerf_hip.h these struct mirrors were written against

c_float_p = C.POINTER(C.c_float)
c_u8_p = C.POINTER(C.c_uint8)
c_i32_p = C.POINTER(C.c_int32)
c_u32_p = C.POINTER(C.c_uint32)
c_f64_p = C.POINTER(C.c_double)

BOX_DOUBLES = 31
SEG_REC_FLOATS = 16     # OBJNERF_SEG_REC_FLOATS
RAY_BIAS_FLOATS = 448   # OBJNERF_RAY_BIAS_FLOATS


class VoxelGrid(C.Structure):
    _fields_ = [
        ("idx_map", C.c_void_p),
        ("table", C.c_void_p),
        ("shape", C.c_int32 * 3),
        ("offset", C.c_float * 3),
        ("voxel_size", C.c_float),
        ("n_rows", C.c_int32),
    ]


class MlpArgs(C.Structure):
    _fields_ = [
        ("use_voxel", C.c_int32), ("do_scene", C.c_int32), ("do_object", C.c_int32),
        ("blob", C.c_void_p), ("aux", C.c_void_p),
        ("rays", C.c_void_p), ("z_vals", C.c_void_p), ("n_rays", C.c_int64), ("S", C.c_int32),
        ("codes", C.c_void_p), ("code_stride", C.c_int64),
        ("grid", VoxelGrid),
        ("emb_xyz", C.c_void_p), ("emb_dir", C.c_void_p), ("obj_voxel", C.c_void_p), ("obj_code", C.c_void_p),
        ("n_points", C.c_int64),
        ("sigma", C.c_void_p), ("rgb", C.c_void_p), ("inst_sigma", C.c_void_p), ("inst_rgb", C.c_void_p),
        ("sigma_only", C.c_int32),
        ("ray_index", C.c_void_p), ("n_active", C.c_void_p),
        ("comp_w", C.c_void_p), ("comp_rec", C.c_void_p), ("comp_last_delta", C.c_float), ("comp_inst_weights", C.c_int32),
        ("ray_bias", C.c_void_p),
        ("points", C.c_void_p), ("lat_x", C.c_void_p), ("lat_y", C.c_void_p), ("lat_z", C.c_void_p), ("lat_n", C.c_int32 * 3),
        ("_pad_lat", C.c_int32),
    ]


class Arch(C.Structure):
    _fields_ = [
        ("D", C.c_int32), ("W", C.c_int32), ("n_skips", C.c_int32), ("skips", C.c_int32 * 8),
        ("inst_D", C.c_int32), ("inst_W", C.c_int32), ("n_inst_skips", C.c_int32), ("inst_skips", C.c_int32 * 8),
        ("in_xyz", C.c_int32), ("in_dir", C.c_int32), ("obj_voxel_c", C.c_int32), ("code_c", C.c_int32),
    ]


class MlpGenericArgs(C.Structure):
    _fields_ = [
        ("arch", Arch), ("h_params", C.POINTER(C.c_void_p)),
        ("do_scene", C.c_int32), ("do_object", C.c_int32), ("sigma_only", C.c_int32), ("_pad", C.c_int32),
        ("n_points", C.c_int64),
        ("emb_xyz", C.c_void_p), ("emb_dir", C.c_void_p), ("obj_voxel", C.c_void_p), ("obj_code", C.c_void_p),
        ("sigma", C.c_void_p), ("rgb", C.c_void_p), ("inst_sigma", C.c_void_p), ("inst_rgb", C.c_void_p),
        ("workspace", C.c_void_p),
    ]


class CompositeArgs(C.Structure):
    _fields_ = [
        ("n_rays", C.c_int64), ("S", C.c_int32),
        ("z_vals", C.c_void_p), ("sigma", C.c_void_p), ("rgb", C.c_void_p),
        ("inst_sigma", C.c_void_p), ("inst_rgb", C.c_void_p),
        ("noise", C.c_void_p), ("noise_inst", C.c_void_p), ("noise_std", C.c_float),
        ("white_back", C.c_int32), ("use_zero_as_last_delta", C.c_int32), ("occlusion", C.c_int32),
        ("frustum_bound_th", C.c_float), ("pass_through_mask", C.c_void_p), ("rays_in_bbox", C.c_int32),
        ("weights", C.c_void_p), ("opacity", C.c_void_p), ("rgb_map", C.c_void_p), ("depth", C.c_void_p),
        ("rgb_inst", C.c_void_p), ("depth_inst", C.c_void_p), ("opacity_inst", C.c_void_p),
    ]


class TrainArgs(C.Structure):
    _fields_ = [
        ("use_voxel", C.c_int32), ("do_object", C.c_int32), ("n_points", C.c_int64),
        ("h_params", C.POINTER(C.c_void_p)),
        ("emb_xyz", C.c_void_p), ("emb_dir", C.c_void_p), ("obj_voxel", C.c_void_p), ("obj_code", C.c_void_p),
        ("sigma", C.c_void_p), ("rgb", C.c_void_p), ("inst_sigma", C.c_void_p), ("inst_rgb", C.c_void_p),
        ("workspace", C.c_void_p),
        ("blob", C.c_void_p), ("aux", C.c_void_p), ("blob_bwd", C.c_void_p),
        ("rays", C.c_void_p), ("z_vals", C.c_void_p), ("n_rays", C.c_int64), ("S", C.c_int32), ("bwd_dx", C.c_int32),
        ("codes", C.c_void_p), ("code_stride", C.c_int64),
        ("grid", VoxelGrid),
        ("scatter_xyz", C.c_void_p), ("scatter_table_grad", C.c_void_p),
        ("emb_dir_ray", C.c_void_p), ("ray_bias_ws", C.c_void_p),
    ]


class CompositeMultiArgs(C.Structure):
    _fields_ = [
        ("n_rays", C.c_int64), ("K", C.c_int32), ("S", C.c_int32),
        ("h_z", C.POINTER(C.c_void_p)), ("h_sigma", C.POINTER(C.c_void_p)), ("h_rgb", C.POINTER(C.c_void_p)),
        ("noise", C.c_void_p), ("noise_std", C.c_float), ("white_back", C.c_int32),
        ("z_sorted", C.c_void_p), ("weights", C.c_void_p), ("obj_ids", C.c_void_p),
        ("opacity", C.c_void_p), ("rgb_map", C.c_void_p), ("depth", C.c_void_p),
        ("h_own_weights", C.POINTER(C.c_void_p)),
        ("scratch", C.c_void_p),
    ]


class RenderCfg(C.Structure):
    _fields_ = [
        ("use_voxel", C.c_int32), ("N_samples", C.c_int32), ("N_importance", C.c_int32), ("use_disp", C.c_int32),
        ("perturb", C.c_float), ("noise_std", C.c_float), ("white_back", C.c_int32),
        ("forward_instance", C.c_int32), ("is_eval", C.c_int32), ("use_zero_as_last_delta", C.c_int32),
        ("frustum_bound_th", C.c_float), ("rays_in_bbox", C.c_int32),
        ("separate_composite", C.c_int32), ("no_hoist", C.c_int32),
    ]


class RenderOut(C.Structure):
    _fields_ = [
        ("weights", C.c_void_p), ("z_vals", C.c_void_p),
        ("opacity", C.c_void_p), ("depth", C.c_void_p), ("depth_instance", C.c_void_p),
        ("opacity_instance", C.c_void_p), ("rgb", C.c_void_p), ("rgb_instance", C.c_void_p),
    ]


class RenderIn(C.Structure):
    _fields_ = [
        ("rays", C.c_void_p), ("n_rays", C.c_int64), ("codes", C.c_void_p), ("code_stride", C.c_int64),
        ("pass_through_mask", C.c_void_p),
        ("blob_coarse", C.c_void_p), ("aux_coarse", C.c_void_p), ("blob_fine", C.c_void_p), ("aux_fine", C.c_void_p),
        ("grid", VoxelGrid),
        ("z_steps", C.c_void_p), ("u_det", C.c_void_p),
        ("perturb_rand", C.c_void_p), ("u_rand", C.c_void_p), ("noise", C.c_void_p * 4),
        ("workspace", C.c_void_p),
    ]


class RenderMultiCfg(C.Structure):
    _fields_ = [
        ("use_voxel", C.c_int32), ("N_samples", C.c_int32), ("N_importance", C.c_int32), ("use_disp", C.c_int32),
        ("perturb", C.c_float), ("noise_std", C.c_float), ("white_back", C.c_int32),
        ("no_hoist", C.c_int32),
    ]


class RenderMultiIn(C.Structure):
    _fields_ = [
        ("n_rays", C.c_int64), ("K", C.c_int32),
        ("h_rays", C.POINTER(C.c_void_p)), ("h_obj_ids", C.POINTER(C.c_int32)), ("code_table", C.c_void_p),
        ("blob_coarse", C.c_void_p), ("aux_coarse", C.c_void_p), ("blob_fine", C.c_void_p), ("aux_fine", C.c_void_p),
        ("grid", VoxelGrid),
        ("z_steps", C.c_void_p), ("u_det", C.c_void_p), ("u_rand", C.c_void_p),
        ("noise_coarse", C.c_void_p), ("noise_fine", C.c_void_p),
        ("h_clip", C.POINTER(C.c_void_p)),
        ("boxes", C.c_void_p), ("n_boxes", C.c_int32),
        ("workspace", C.c_void_p),
    ]


class RenderMultiOut(C.Structure):
    _fields_ = [
        ("z_vals", C.c_void_p), ("weights", C.c_void_p), ("obj_ids", C.c_void_p),
        ("opacity", C.c_void_p), ("depth", C.c_void_p), ("rgb", C.c_void_p),
    ]


# every symbol include/objnerf_hip.h declares: (restype, argtypes)
_VP = C.c_void_p
SIGNATURES = {
    "objnerf_abi_version": (C.c_int, []),
    "objnerf_last_error": (C.c_char_p, []),
    "objnerf_blob_floats": (C.c_int64, [C.c_int]),
    "objnerf_aux_floats": (C.c_int64, []),
    "objnerf_num_param_ptrs": (C.c_int, []),
    "objnerf_param_numel": (C.c_int64, [C.c_int, C.c_int]),
    "objnerf_pack_index": (C.c_int, [C.c_int, _VP, _VP]),
    "objnerf_pack_weights": (C.c_int, [C.c_int, _VP, _VP, C.POINTER(_VP), _VP, _VP, _VP]),
    "objnerf_pack_models": (C.c_int, [C.c_int, _VP, _VP, C.c_int, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP), _VP]),
    "objnerf_bwd_blob_floats": (C.c_int64, []),
    "objnerf_pack_index_bwd": (C.c_int, [C.c_int, _VP]),
    "objnerf_pack_weights_bwd": (C.c_int, [_VP, C.POINTER(_VP), _VP, _VP]),
    "objnerf_sample_coarse": (C.c_int, [_VP, _VP, _VP, C.c_float, C.c_int, C.c_int64, C.c_int, _VP, _VP]),
    "objnerf_pos_encode": (C.c_int, [_VP, C.c_int64, C.c_int, C.c_int, _VP, _VP]),
    "objnerf_pos_encode_freqs": (C.c_int, [_VP, C.c_int64, C.c_int, C.c_int, _VP, _VP, _VP]),
    "objnerf_voxel_embed": (C.c_int, [C.POINTER(VoxelGrid), _VP, C.c_int64, _VP, _VP, _VP]),
    "objnerf_mlp_eval": (C.c_int, [C.POINTER(MlpArgs), _VP]),
    "objnerf_ray_bias": (C.c_int, [C.POINTER(MlpArgs), _VP, _VP]),
    "objnerf_ray_bias_floats": (C.c_int64, [C.c_int64]),
    "objnerf_composite": (C.c_int, [C.POINTER(CompositeArgs), _VP]),
    "objnerf_composite_finish": (C.c_int, [_VP, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "objnerf_sample_pdf_merge": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_float, _VP, _VP, _VP]),
    "objnerf_sample_pdf_merge_clip": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_float, _VP, _VP, _VP, _VP]),
    "objnerf_sample_pdf": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_float, _VP, _VP]),
    "objnerf_mask_sigma": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_int, _VP, C.c_int, _VP]),
    "objnerf_mask_sigma_rgb": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, C.c_int, _VP, C.c_int, _VP]),
    "objnerf_compact_scratch_ints": (C.c_int64, [C.c_int64]),
    "objnerf_compact_rays": (C.c_int, [_VP, C.c_int64, C.c_int, _VP, _VP, _VP, _VP]),
    "objnerf_points_in_boxes": (C.c_int, [_VP, C.c_int64, _VP, C.c_int, _VP, _VP]),
    "objnerf_composite_multi": (C.c_int, [C.POINTER(CompositeMultiArgs), _VP]),
    "objnerf_composite_multi_scratch_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "objnerf_generate_rays": (C.c_int, [C.c_int, C.c_int, C.c_float, _VP, C.c_float, C.c_float, _VP, C.c_double, _VP, _VP]),
    "objnerf_generate_rays_rows": (C.c_int, [C.c_int, C.c_int, C.c_float, _VP, C.c_float, C.c_float, _VP, C.c_double,
                                             C.c_int, C.c_int, C.c_int, C.c_int, _VP, _VP]),
    "objnerf_ray_directions": (C.c_int, [C.c_int, C.c_int, C.c_float, _VP, _VP]),
    "objnerf_get_rays": (C.c_int, [_VP, C.c_int64, _VP, C.c_int, _VP, _VP, _VP]),
    "objnerf_ray_box_near_far": (C.c_int, [_VP, _VP, C.c_int64, _VP, C.c_double, _VP, _VP, _VP, _VP]),
    "objnerf_render_workspace_bytes": (C.c_int64, [C.POINTER(RenderCfg), C.c_int64]),
    "objnerf_render_rays": (C.c_int, [C.POINTER(RenderCfg), C.POINTER(RenderIn), C.POINTER(RenderOut), C.POINTER(RenderOut), _VP]),
    "objnerf_render_multi_workspace_bytes": (C.c_int64, [C.POINTER(RenderMultiCfg), C.c_int32, C.c_int64]),
    "objnerf_render_rays_multi": (C.c_int, [C.POINTER(RenderMultiCfg), C.POINTER(RenderMultiIn), C.POINTER(RenderMultiOut),
                                            C.POINTER(RenderMultiOut), _VP]),
    "objnerf_gemm": (C.c_int, [_VP, C.c_int64, C.c_int, _VP, C.c_int64, C.c_int, _VP, C.c_int64, C.c_int64, C.c_int64,
                               C.c_int64, C.c_int, C.c_int, _VP, C.c_int, _VP]),
    "objnerf_train_workspace_floats": (C.c_int64, [C.c_int, C.c_int64]),
    "objnerf_train_scratch_floats": (C.c_int64, [C.c_int64]),
    "objnerf_mlp_train_forward": (C.c_int, [C.POINTER(TrainArgs), _VP]),
    "objnerf_mlp_train_backward": (C.c_int, [C.POINTER(TrainArgs), _VP, _VP, _VP, _VP, C.POINTER(_VP), _VP, _VP, _VP, _VP, _VP]),
    "objnerf_composite_backward": (C.c_int, [C.POINTER(CompositeArgs), _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "objnerf_voxel_embed_backward": (C.c_int, [C.POINTER(VoxelGrid), _VP, C.c_int64, _VP, _VP, _VP, _VP]),
    "objnerf_sum_over_samples": (C.c_int, [_VP, C.c_int64, C.c_int, C.c_int, _VP, _VP]),
    "objnerf_rows_gather_backward": (C.c_int, [_VP, _VP, C.c_int64, C.c_int, C.c_int64, _VP, _VP]),
    "objnerf_sample_points": (C.c_int, [_VP, _VP, C.c_int64, C.c_int, _VP, _VP]),
    "objnerf_arch_num_param_ptrs": (C.c_int, [C.POINTER(Arch)]),
    "objnerf_mlp_generic_workspace_floats": (C.c_int64, [C.POINTER(Arch), C.c_int64]),
    "objnerf_mlp_generic": (C.c_int, [C.POINTER(MlpGenericArgs), _VP]),
    "objnerf_mlp_generic_train_workspace_floats": (C.c_int64, [C.POINTER(Arch), C.c_int64]),
    "objnerf_mlp_generic_train_scratch_floats": (C.c_int64, [C.POINTER(Arch), C.c_int64]),
    "objnerf_mlp_generic_train_forward": (C.c_int, [C.POINTER(MlpGenericArgs), _VP]),
    "objnerf_mlp_generic_train_backward": (C.c_int, [C.POINTER(MlpGenericArgs), _VP, _VP, _VP, _VP, C.POINTER(_VP), _VP, C.c_int, _VP, _VP, _VP, _VP]),
    "objnerf_pos_encode_block_backward": (C.c_int, [_VP, C.c_int64, C.c_int64, C.c_int, C.c_int, _VP, _VP, C.c_int64, _VP, C.c_int64, _VP]),
    "objnerf_voxel_features_backward": (C.c_int, [C.POINTER(VoxelGrid), C.c_int, _VP, C.c_int64, _VP, C.c_int64, _VP, _VP]),
    "objnerf_voxel_features": (C.c_int, [C.POINTER(VoxelGrid), C.c_int, _VP, C.c_int64, _VP, C.c_int64, _VP]),
    "objnerf_pos_encode_block": (C.c_int, [_VP, C.c_int64, C.c_int64, C.c_int, C.c_int, _VP, _VP, C.c_int64, _VP]),
    "objnerf_repeat_rows": (C.c_int, [_VP, C.c_int64, C.c_int64, C.c_int, C.c_int, _VP, C.c_int64, _VP]),
    "objnerf_timing_enable": (C.c_int, [C.c_int]),
    "objnerf_timing_read": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "objnerf_train_timing_enable": (C.c_int, [C.c_int]),
    "objnerf_train_timing_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

_lib = None


def lib():
    """Loads the shared library (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                "object_nerf_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C object_nerf_amd/csrc`. There is no fallback path." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the ABI header and the library disagree
            fn.restype = res
            fn.argtypes = args
        if l.objnerf_abi_version() != ABI_VERSION:
            raise RuntimeError("object_nerf_amd: ABI version mismatch (library %d, Python mirror %d): rebuild with "
                               "`make -C object_nerf_amd/csrc`" % (l.objnerf_abi_version(), ABI_VERSION))
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().objnerf_last_error().decode("utf-8", "replace")
        raise RuntimeError("objnerf_hip %s failed (rc=%d): %s" % (what, rc, msg))


def stream_ptr():
    """torch's current stream on the CURRENT device; entry points run under `on_device_of`, which makes the
    tensors' device current first (the library sizes its grids from hipGetDevice())."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def on_device_of(pick):
    """Decorator for the public entry points: run the call with the device of `pick(*args, **kwargs)` (a tensor)
    current, so that rays / models on cuda:1 work while cuda:0 is the current device, as they do with the
    PyTorch reference (launch stream, grid sizing and allocations all follow the current device)."""
    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            try:
                t = pick(*args, **kwargs)
            except (IndexError, KeyError, TypeError):
                t = None
            if isinstance(t, torch.Tensor) and t.is_cuda and t.device.index != torch.cuda.current_device():
                with torch.cuda.device(t.device):
                    return fn(*args, **kwargs)
            return fn(*args, **kwargs)
        return wrapper
    return deco


def ptr(t):
    """Device pointer of a contiguous tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "objnerf_hip needs contiguous tensors"
    return C.c_void_p(t.data_ptr())


def require_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(
            "object_nerf_amd: %s must live on the GPU (got %s). The HIP path has no CPU fallback." % (name, t.device))


def as_f32(t):
    """contiguous fp32 view/copy of t"""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
