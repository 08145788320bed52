"""CPU, build container only: the oracle against the reference imported LIVE from /root/reference
(skipped where the mount is absent, e.g. on the GPU box).  Bit-exact agreement is asserted: the
oracle issues the same ATen ops in the same order."""
import types

import numpy as np
import pytest
import torch

import cases
import helpers as H
from oracle import ref_import
from oracle import objnerf_oracle as O
from object_nerf_amd.config import AttrDict

pytestmark = pytest.mark.skipif(not ref_import.reference_available(), reason="reference tree not mounted")


@pytest.fixture(scope="module")
def ref():
    return ref_import.load_reference()


def ref_scene(ref, name):
    def mk_ev(ch, nf, mv, conf):
        c = AttrDict(conf)
        key = "live_%d.ply" % len(ref_import.POINT_CLOUDS)
        ref_import.POINT_CLOUDS[key] = np.asarray(conf["pcd_xyz"])
        c["pcd_path"] = key
        return ref.EmbeddingVoxel(ch, nf, mv, c)
    rt = types.SimpleNamespace(ObjectNeRF=ref.ObjectNeRF, Embedding=ref.Embedding, EmbeddingVoxel=mk_ev,
                               CodeLibrary=ref.CodeLibrary)
    return cases.scene_for(rt, name)


@pytest.mark.parametrize("case", ["voxel_eval", "plain_eval", "voxel_train_flags"])
def test_render_rays_bit_exact(ref, case):
    c = cases.RENDER_CASES[case]
    sc = ref_scene(ref, c["scene"])
    rays, ids, ptm, _ = cases.render_inputs(case)
    with torch.no_grad():
        codes = sc.code_library({"instance_ids": ids})["embedding_instance"]
        kw = dict(c["kw"])
        want = ref.render_rays(sc.models, sc.embeddings, rays, perturb=0, noise_std=0, embedding_instance=codes,
                               pass_through_mask=ptm, chunk=4096, **kw)
        grid = H.oracle_grid(sc.embeddings["xyz"]) if cases.SCENES[c["scene"]][0] else None
        got = O.render_rays(H.state(sc.models["coarse"]), H.state(sc.models["fine"]), grid, rays,
                            embedding_instance=codes, pass_through_mask=ptm, chunk=4096, **kw)
    assert sorted(got) == sorted(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k


def test_drop_in_state_matches_reference(ref):
    """same seeds -> the drop-in module types hold exactly the reference's parameters and buffers"""
    import object_nerf_amd as A
    a, b = ref_scene(ref, "voxel"), cases.scene_for(A, "voxel")
    for ma, mb in ((a.models["coarse"], b.models["coarse"]), (a.embeddings["xyz"], b.embeddings["xyz"]),
                   (a.code_library, b.code_library)):
        sa, sb = ma.state_dict(), mb.state_dict()
        assert list(sa) == list(sb)
        for k in sa:
            assert sa[k].dtype == sb[k].dtype and torch.equal(sa[k], sb[k]), k
