"""CPU: the C-ABI library loads and exports every symbol include/objnerf_hip.h declares, the
weight-packing index maps are a bijection onto the reference's parameter tensors, and the host-side
plumbing (module types, error behaviour, sharding arithmetic) behaves.  No kernel is launched."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import cases
import helpers as H
import object_nerf_amd as A
from object_nerf_amd import _lib
from object_nerf_amd.distributed import shard_bounds
from object_nerf_amd.nerf_model import PARAM_LAYERS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "objnerf_hip.h")).read()
    declared = set(re.findall(r"\b(objnerf_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "missing export: " + name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    l = _lib.lib()
    version = int(re.search(r"#define OBJNERF_ABI_VERSION (\d+)", hdr).group(1))
    assert l.objnerf_abi_version() == version == _lib.ABI_VERSION
    assert l.objnerf_blob_floats(1) == 111 * 8192 and l.objnerf_blob_floats(0) == 87 * 8192
    assert l.objnerf_num_param_ptrs() == 2 * len(PARAM_LAYERS) == 40


@pytest.mark.parametrize("use_voxel", [True, False])
def test_pack_index_is_a_bijection(use_voxel):
    """every element of every nn.Linear weight/bias appears exactly once in blob+aux; the rest is padding"""
    l = _lib.lib()
    uv = int(use_voxel)
    nb, na = l.objnerf_blob_floats(uv), l.objnerf_aux_floats()
    bi = torch.empty(nb, dtype=torch.int32)
    ai = torch.empty(na, dtype=torch.int32)
    assert l.objnerf_pack_index(uv, C.c_void_p(bi.data_ptr()), C.c_void_p(ai.data_ptr())) == 0
    allidx = torch.cat([bi, ai]).to(torch.int64) & 0xFFFFFFFF
    used = allidx[allidx != 0xFFFFFFFF]
    m = A.ObjectNeRF(A.default_model_config(use_voxel_embedding=use_voxel))
    params = m._param_list()
    total = 0
    for pid, p in enumerate(params):
        assert p.numel() == l.objnerf_param_numel(uv, pid)
        offs = used[(used >> 24) == pid] & 0xFFFFFF
        assert offs.numel() == p.numel(), (pid, offs.numel(), p.numel())
        assert torch.equal(torch.sort(offs)[0], torch.arange(p.numel()))
        total += p.numel()
    assert used.numel() == total == sum(p.numel() for p in m.parameters())
    # weights live in the stream, biases and heads in the aux block
    assert set(((bi.to(torch.int64) & 0xFFFFFFFF)[bi != -1] >> 24).unique().tolist()) <= set(range(0, 40, 2))


@pytest.mark.parametrize("mode", [1, 0, 2])
def test_backward_weight_stream_references_the_right_elements(mode):
    """backward stream: exactly the hidden-to-hidden blocks, each element once, as a padding-free prefix of the buffer (modes 0 / 1);
    mode 2 (ABI 10, objnerf_train_args.bwd_dx) adds the embedding-column blocks of xyz_encoding_5 / _1 and instance_encoding_3 / _1
    -- 208 scene-voxel and 104 object-voxel columns -- each element once as well, zero padding where a 7-tile layer's chunk has no
    tile (16 of its 128 slots) and in the rows past column 208 / 104 of a block's last tile."""
    l = _lib.lib()
    nbw = l.objnerf_bwd_blob_floats()
    bw = torch.empty(nbw, dtype=torch.int32)
    _lib.check(l.objnerf_pack_index_bwd(mode, C.c_void_p(bw.data_ptr())), "pack_index_bwd")
    w_all = bw.numpy().view("uint32")
    w = w_all[w_all != 0xFFFFFFFF]
    assert np.unique(w).size == w.size                                   # every streamed element exactly once
    # expected element count: SD[:, :256], SF, S8..S6, S5[:, hidden], S4..S2 (256 x 256 each, SD 128 x 256) and
    # OD[:, :128] (64 x 128), OF, O4, O3[:, hidden], O2 (128 x 128 each)
    hidden = 128 * 256 + 8 * 256 * 256 + 64 * 128 + 4 * 128 * 128
    names = [n for n in PARAM_LAYERS]
    want = {"dir_encoding.0", "xyz_encoding_final", "xyz_encoding_8.0", "xyz_encoding_7.0", "xyz_encoding_6.0",
            "xyz_encoding_5.0", "xyz_encoding_4.0", "xyz_encoding_3.0", "xyz_encoding_2.0", "inst_dir_encoding.0",
            "instance_encoding_final.0", "instance_encoding_4.0", "instance_encoding_3.0", "instance_encoding_2.0"}
    if mode < 2:
        assert w.size == hidden
        assert (w_all[:hidden] != 0xFFFFFFFF).all() and (w_all[hidden:] == 0xFFFFFFFF).all()       # whole tiles, then the unused tail
    else:
        assert w.size == hidden + 2 * 208 * 256 + 2 * 208 * 128 + 2 * 104 * 128
        want |= {"xyz_encoding_1.0", "instance_encoding_1.0"}
        # the embedding-column blocks reference exactly columns [0, 208) of the four layers and [271, 375) of the instance layers
        for lname, in_features, cols in (("xyz_encoding_1.0", 271, range(0, 208)), ("instance_encoding_1.0", 439, list(range(0, 208)) + list(range(271, 375)))):
            pid = 2 * names.index(lname)
            offs = (w[(w >> 24) == pid] & 0xFFFFFF).astype(np.int64)
            assert sorted(set((offs % in_features).tolist())) == sorted(cols), lname
        with pytest.raises(RuntimeError):
            _lib.check(l.objnerf_pack_index_bwd(3, C.c_void_p(bw.data_ptr())), "pack_index_bwd")
    # pointer ids: weights only (even ids) of the layers named above
    assert set((w >> 24).tolist()) == {2 * names.index(n) for n in want}


def test_module_types_and_state_dict_names():
    sc = cases.scene_for(A, "voxel")
    m = sc.models["coarse"]
    names = set(m.state_dict())
    for lname in PARAM_LAYERS:
        assert lname + ".weight" in names and lname + ".bias" in names
    assert m.in_channels_xyz == 271 and m.inst_channel_in == 439 and m.in_channels_dir == 27
    assert sum(p.numel() for p in m.parameters()) == 891_208          # SURVEY.md §2.3 [probed]
    ev = sc.embeddings["xyz"]
    for b in ("voxel_size", "bounds", "voxel_offset", "voxel_shape", "voxel_count", "voxel_occupancy", "voxel_idx_map"):
        assert b in dict(ev.named_buffers())
    assert ev.voxel_idx_map.dtype == torch.int64 and ev.embedding_space_ftr.weight.shape == (cases.MAX_VOXELS, 24)
    assert sc.code_library({"instance_ids": torch.tensor([[1], [3]])})["embedding_instance"].shape == (2, 64)
    assert sc.code_library({}) == {}
    assert A.Embedding(3, 10).out_channels == 63 and A.Embedding(3, 4).out_channels == 27
    plain = A.ObjectNeRF(A.default_model_config(use_voxel_embedding=False))
    assert plain.in_channels_xyz == 63 and plain.inst_channel_in == 127
    assert sum(p.numel() for p in plain.parameters()) == 704_840


def test_other_architectures_are_built_like_the_reference_builds_them():
    """models/nerf_model.py:18-95 builds any config.model shape; so does the drop-in (round 3 raised for anything but the
    shipped default).  Non-default shapes are flagged for the layer-wise path; parameter names and shapes follow the config."""
    assert A.ObjectNeRF(A.default_model_config()).fused_architecture
    m = A.ObjectNeRF(A.default_model_config(W=128, D=6, skips=[3], inst_D=3, inst_W=64, inst_skips=[1], N_freq_xyz=8,
                                            N_freq_dir=3, N_obj_code_length=32, use_voxel_embedding=False))
    assert not m.fused_architecture and m.in_channels_xyz == 51 and m.in_channels_dir == 21 and m.inst_channel_in == 83
    sd = m.state_dict()
    assert sd["xyz_encoding_4.0.weight"].shape == (128, 128 + 51) and sd["xyz_encoding_6.0.weight"].shape == (128, 128)
    assert "xyz_encoding_7.0.weight" not in sd and sd["instance_encoding_2.0.weight"].shape == (64, 64 + 83)
    assert sd["dir_encoding.0.weight"].shape == (64, 128 + 21) and sd["inst_rgb.0.weight"].shape == (3, 32)
    from object_nerf_amd import generic
    a = generic.arch_of(m)
    assert (a.D, a.W, a.n_skips, a.skips[0], a.inst_D, a.inst_W, a.inst_skips[0], a.in_xyz, a.in_dir, a.obj_voxel_c, a.code_c) == \
        (6, 128, 1, 3, 3, 64, 1, 51, 21, 0, 32)
    l = _lib.lib()
    assert l.objnerf_arch_num_param_ptrs(C.byref(a)) == 2 * (6 + 4) + 2 * (3 + 4)
    # two ping-pong hidden buffers | final | direction hidden per point, then (round 6) a constant: the packed weight stream of the
    # longest thing one persistent kernel may take -- a run of plain layers (6 x (2 chunks of 8,192 floats for a 128-wide layer + 256
    # bias floats)) or the whole 128-wide scene branch (17 chunks + biases + heads + its description); the 64-wide object branch stays on the
    # GEMMs (csrc/chain_generic.hip: widths 96 .. 256)
    ws0, ws1 = (l.objnerf_mlp_generic_workspace_floats(C.byref(a), n) for n in (0, 1000))
    # (a 128-wide branch takes its 51 input columns as ONE 64-column block: 2 block chunks + 12 up to `final` + 1 direction-embedding
    # block + 2 chunks of the direction layer's hidden columns)
    assert ws1 - ws0 == 1000 * (2 * 128 + 128 + 64) and 17 * 8192 < ws0 < 17 * 8192 + 6144
    a.W = 127
    assert l.objnerf_mlp_generic_workspace_floats(C.byref(a), 1000) < 0 and b"architecture" in l.objnerf_last_error()
    with pytest.raises(ValueError):
        A.ObjectNeRF(A.default_model_config(W=255))
    e = A.Embedding(3, 4, logscale=False)           # built (frequency table); only the fused renderer insists on 2^k bands
    assert torch.equal(e.freq_bands, torch.linspace(1, 8, 4)) and e.out_channels == 27


def test_no_silent_cpu_fallback():
    """CPU tensors must raise, not compute: the HIP path is the only path"""
    sc = cases.scene_for(A, "plain")
    rays = H.test_rays(8)
    codes = torch.zeros(8, 64)
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="GPU"):
            A.render_rays(sc.models, sc.embeddings, rays, N_samples=8, embedding_instance=codes, noise_std=0)
        with pytest.raises(RuntimeError, match="GPU"):
            sc.models["coarse"]({"emb_xyz": torch.zeros(4, 63), "emb_dir": torch.zeros(4, 27)})
        with pytest.raises(RuntimeError, match="GPU"):
            A.Embedding(3, 4)(torch.zeros(4, 3))
        with pytest.raises(TypeError):
            A.render_rays(sc.models, sc.embeddings, rays, N_samples=8, noise_std=0)      # embedding_instance missing
    with pytest.raises(RuntimeError, match="GPU"):      # grad mode selects the HIP training path: still no CPU fallback
        A.render_rays(sc.models, sc.embeddings, rays, N_samples=8, embedding_instance=codes, noise_std=0)


def test_attrdict_access_forms():
    c = A.default_model_config()
    assert c.D == c["D"] == c.get("D", 0) == 8 and c.get("missing", 7) == 7
    with pytest.raises(AttributeError):
        c.missing


def test_shard_bounds_cover_all_rays():
    for n in (0, 1, 7, 307200, 307201):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a0 <= a1
            assert max(hi - lo for lo, hi in spans) == (n + world - 1) // world


def test_c_abi_argument_validation():
    """entry points reject bad arguments with an error string instead of launching"""
    l = _lib.lib()
    a = _lib.MlpArgs()
    assert l.objnerf_mlp_eval(C.byref(a), None) < 0
    assert b"null weights" in l.objnerf_last_error()
    assert l.objnerf_sample_coarse(None, None, None, 0.0, 0, 4, 8, None, None) < 0
    assert l.objnerf_sample_pdf_merge(None, None, None, 0, 1, 64, 64, 1e-5, None, None, None) < 0
    # workspace: sigma / rgb of both branches (32 B per sample) in the two-kernel form; only the 64-byte segment records
    # per 32 samples when the passes composite in the MLP kernel's epilogue (no occlusion mask, no noise, S % 32 == 0);
    # plus, in fp32 arithmetic, the per-ray vectors of the hoisted terms (1792 B per ray) unless no_hoist
    rb = _lib.RAY_BIAS_FLOATS
    rb_total = l.objnerf_ray_bias_floats(1000)                    # the vectors (the hoisted weight columns sit behind aux since round 4;
    # round 6: as the A-operand stream of the MFMA ray_bias kernel -- 8 tiles x 8 + 6 tiles x 4 groups of 256 floats -- + 448 biases)
    assert rb_total == 1000 * rb and l.objnerf_aux_floats() == 4112 + (8 * 8 + 6 * 4) * 256 + 448
    cfg = _lib.RenderCfg(N_samples=64, N_importance=64, separate_composite=1, no_hoist=1)
    assert l.objnerf_render_workspace_bytes(C.byref(cfg), 1000) == 4 * 1000 * 128 * 8 + 256
    cfg = _lib.RenderCfg(N_samples=64, N_importance=64, no_hoist=1)
    assert l.objnerf_render_workspace_bytes(C.byref(cfg), 1000) == 4 * 1000 * (128 // 32) * _lib.SEG_REC_FLOATS + 256
    cfg = _lib.RenderCfg(N_samples=64, N_importance=64)
    assert l.objnerf_render_workspace_bytes(C.byref(cfg), 1000) == 4 * (1000 * (128 // 32) * _lib.SEG_REC_FLOATS + rb_total) + 256
    for kw in (dict(noise_std=1.0), dict(is_eval=0, frustum_bound_th=0.025)):        # noise / occlusion mask: two-kernel form
        c2 = _lib.RenderCfg(N_samples=64, N_importance=64, no_hoist=1, **kw)
        assert l.objnerf_render_workspace_bytes(C.byref(c2), 1000) == 4 * 1000 * 128 * 8 + 256
    # odd coarse count, fine count a multiple of 32: the coarse pass needs the two-kernel form, the fine pass does not
    c3 = _lib.RenderCfg(N_samples=40, N_importance=24, no_hoist=1)
    assert l.objnerf_render_workspace_bytes(C.byref(c3), 1000) == 4 * 1000 * 40 * 8 + 256
    # the multi-object entry points (round 2)
    assert l.objnerf_compact_rays(None, 10, 4, None, None, None, None) < 0 and b"compact_rays" in l.objnerf_last_error()
    assert l.objnerf_compact_scratch_ints(5000) == 5 + 1
    assert l.objnerf_sample_pdf_merge_clip(None, None, None, 0, 1, 64, 64, 1e-5, None, None, None, None) < 0
    mc = _lib.RenderMultiCfg(N_samples=64, N_importance=64, no_hoist=1)
    per_set = (1000 * (64 + 128 + 4 * 128 + 64) + 1000 + 15) // 16 * 16       # depths, sigma / rgb, own weights, ray index; 64-byte units
    assert l.objnerf_render_multi_workspace_bytes(C.byref(mc), 3, 1000) == 4 * (3 * per_set + 64 * 3 + 64) + 256
    mc = _lib.RenderMultiCfg(N_samples=64, N_importance=64)                    # + the per-ray vectors of each set (fp32 passes)
    assert l.objnerf_render_multi_workspace_bytes(C.byref(mc), 3, 1000) == 4 * (3 * (per_set + (l.objnerf_ray_bias_floats(1000) + 15) // 16 * 16) + 64 * 3 + 64) + 256
    # the joint compositing stages K * (S + I) samples of 28 bytes: in LDS up to 152 KiB, in the workspace beyond (1024 slices)
    assert l.objnerf_composite_multi_scratch_bytes(3, 128) == 0 and l.objnerf_composite_multi_scratch_bytes(20, 128) == 0
    assert l.objnerf_composite_multi_scratch_bytes(3, 2000) == 1024 * 6000 * 28
    big = _lib.RenderMultiCfg(N_samples=1000, N_importance=1000, no_hoist=1)
    small = _lib.RenderMultiCfg(N_samples=64, N_importance=64, no_hoist=1)
    grow = l.objnerf_render_multi_workspace_bytes(C.byref(big), 3, 8) - 1024 * 6000 * 28
    assert 0 < grow < 4 * 3 * 8 * (1000 + 2000 + 4 * 2000 + 1000 + 1 + 16) + 4 * (64 * 3 + 64) + 256 + 64
    assert l.objnerf_render_multi_workspace_bytes(C.byref(small), 3, 8) < grow
    rin, out = _lib.RenderMultiIn(), _lib.RenderMultiOut()
    rin.n_rays, rin.K = 8, 0
    assert l.objnerf_render_rays_multi(C.byref(mc), C.byref(rin), C.byref(out), None, None) < 0
    assert b"bad sizes" in l.objnerf_last_error()
    rin.K = 2
    assert l.objnerf_render_rays_multi(C.byref(mc), C.byref(rin), C.byref(out), None, None) < 0
    assert b"missing input" in l.objnerf_last_error()
    # round 3: the limits of later stages are checked BEFORE the first launch.  Round 4: the joint compositing
    # has no sample limit any more (13 x 192 samples no longer refused); K <= 64 and the sampler's per-set sizes remain
    rin.K = 65
    assert l.objnerf_render_rays_multi(C.byref(mc), C.byref(rin), C.byref(out), None, None) < 0
    assert b"bad sizes" in l.objnerf_last_error()
    rin.K = 2
    wide = _lib.RenderMultiCfg(N_samples=1500, N_importance=1000)
    assert l.objnerf_render_rays_multi(C.byref(wide), C.byref(rin), C.byref(out), None, None) < 0
    assert b"importance sampler" in l.objnerf_last_error()
    tiny = _lib.RenderMultiCfg(N_samples=2, N_importance=4)
    assert l.objnerf_render_rays_multi(C.byref(tiny), C.byref(rin), C.byref(out), None, None) < 0
    assert b"N_samples >= 3" in l.objnerf_last_error()
    # ray subset of objnerf_mlp_eval: list and count go together, fused form only
    a = _lib.MlpArgs()
    a.blob = a.aux = 64                                           # non-null dummies: validation never dereferences them
    a.rays = a.z_vals = a.sigma = 64
    a.n_rays, a.S, a.do_scene = 4, 8, 1
    a.ray_index = 64
    assert l.objnerf_mlp_eval(C.byref(a), None) < 0 and b"go together" in l.objnerf_last_error()
    a.ray_index, a.n_active = None, 64
    assert l.objnerf_mlp_eval(C.byref(a), None) < 0 and b"go together" in l.objnerf_last_error()
    a.ray_index = 64
    a.emb_xyz = a.emb_dir = 64                                    # memory form
    a.n_points = 4
    assert l.objnerf_mlp_eval(C.byref(a), None) < 0 and b"fused form" in l.objnerf_last_error()
    # compositing in the epilogue: fused form, scene branch, records, S % 32 == 0, no ray subset
    b = _lib.MlpArgs()
    b.blob = b.aux = b.rays = b.z_vals = 64
    b.n_rays, b.S, b.do_scene, b.comp_w = 4, 40, 1, 64
    assert l.objnerf_mlp_eval(C.byref(b), None) < 0 and b"comp_w needs" in l.objnerf_last_error()
    b.S, b.comp_rec, b.do_scene, b.do_object = 64, 64, 0, 1
    assert l.objnerf_mlp_eval(C.byref(b), None) < 0 and b"comp_w needs" in l.objnerf_last_error()
    b.do_scene, b.do_object, b.comp_w, b.comp_rec, b.sigma = 1, 0, None, None, 64
    b.ray_bias, b.emb_xyz, b.emb_dir, b.n_points = 64, 64, 64, 4                  # memory form: no rays to hoist over
    assert l.objnerf_mlp_eval(C.byref(b), None) < 0 and b"ray_bias needs" in l.objnerf_last_error()
    assert l.objnerf_ray_bias(C.byref(b), None, None) < 0 and b"ray_bias: bad arguments" in l.objnerf_last_error()
    assert l.objnerf_composite_finish(None, 4, 40, 0, 0, 0, None, None, None, None, None, None, None, None) < 0
    assert b"multiple of 32" in l.objnerf_last_error()


def test_library_never_allocates_or_synchronises():
    """include/objnerf_hip.h's contract: entry points only ENQUEUE on the caller's stream.  The permitted waits are
    objnerf_timing_read / objnerf_train_timing_read (bench.py's measurement hooks) on their own events; there is no device allocation, no blocking copy
    and no stream / device synchronisation anywhere under csrc/ -- in particular none inside objnerf_render_rays_multi,
    whose ray culling counts on the device (objnerf_compact_rays) instead of reading a count back."""
    import glob
    import re
    csrc = os.path.join(ROOT, "object_nerf_amd", "csrc")
    banned = re.compile(r"\b(hipMalloc\w*|hipFree\w*|hipMemcpy(?!Async)\w*|hipDeviceSynchronize|hipStreamSynchronize|"
                        r"hipStreamWaitEvent|hipHostMalloc|hipMemcpyAsync)\b")
    hits = []
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        for no, line in enumerate(open(path), 1):
            code = line.split("//")[0]
            if banned.search(code):
                hits.append("%s:%d %s" % (os.path.basename(path), no, code.strip()))
    assert not hits, hits
    waits = {}
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        n = sum("hipEventSynchronize" in ln.split("//")[0] for ln in open(path))
        if n:
            waits[os.path.basename(path)] = n
    assert waits == {"api.hip": 1, "train.hip": 1}        # objnerf_timing_read, objnerf_train_timing_read


def test_weight_stream_is_not_cached():
    """round 6: the packed weight stream is gathered from the parameters at every call -- there is no cache key to go stale (rounds
    1-5 keyed on (data_ptr, _version) + an optimizer-step hook and kept finding writers the key could not see: fused optimizers,
    `.data` writes, graph-replayed steps, deep copies carrying the original's parameter ids).  A module therefore carries no
    cache state at all, registers no optimizer hook, and a deep copy / pickle round trip resolves ITS OWN parameters."""
    import copy
    import pickle
    from object_nerf_amd import nerf_model
    m = A.ObjectNeRF(A.default_model_config())
    assert not [k for k in vars(m) if "pack" in k or "epoch" in k or "param_id" in k]
    assert not hasattr(nerf_model, "_on_optimizer_step") and not hasattr(nerf_model, "_live_models")
    m.invalidate_packed()                                     # kept as a no-op for rounds-1-5 callers
    for c in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        pl, own = c._param_list(), {id(p) for p in c.parameters()}
        assert len(pl) == _lib.lib().objnerf_num_param_ptrs() and all(id(p) in own for p in pl)
        assert not any(id(p) in own for p in m._param_list())
    # replacing a parameter object (not just its values) is seen too: the list is resolved per call, not remembered
    m.sigma.weight = torch.nn.Parameter(torch.zeros_like(m.sigma.weight))
    assert any(p is m.sigma.weight for p in m._param_list())


def test_training_workspace_sizes():
    """size queries of the training calls (the library never allocates): the activation workspace = 12.6 KB of saved layer outputs
    per point (2,436 floats scene + 708 object) + the LeakyReLU sign masks the fused forward packs (14 groups x 64 lanes x 16
    bytes per 32 points, whole 128-point tiles); the backward's scratch = the gradient matrices in the same layout + the partial
    tiles of the grouped weight-gradient pass + the per-ray terms' area -- monotone in the point count and 16-byte granular"""
    l = _lib.lib()
    for P in (1, 127, 128, 129, 4096, 131072, 262144):
        masks = ((P + 127) // 128) * 4 * 14 * 256
        assert l.objnerf_train_workspace_floats(1, P) == (2436 + 708) * P + masks
        assert l.objnerf_train_workspace_floats(0, P) == 2436 * P + masks
    prev = 0
    for P in (16, 384, 768, 131072, 262144, 393216):
        s = l.objnerf_train_scratch_floats(P)
        assert s > (2436 + 708) * P + ((P + 15) // 16) * (448 + 92) and s % 4 == 0 and s > prev
        prev = s
    assert l.objnerf_train_timing_enable(0) == 0
    ms, cnt = (C.c_double * 5)(), (C.c_int64 * 5)()
    assert l.objnerf_train_timing_read(ms, cnt) == 0 and list(cnt) == [0] * 5 and list(ms) == [0.0] * 5
    # the row-gather backward validates before it launches
    assert l.objnerf_rows_gather_backward(None, None, 4, 64, 64, None, None) < 0 and b"rows_gather_backward" in l.objnerf_last_error()
    assert l.objnerf_rows_gather_backward(64, 64, 4, 2048, 64, 64, None) < 0
