"""CPU: Lightning-style checkpoint round trip through the drop-in types."""
import io

import pytest
import torch

import cases
import object_nerf_amd as A
from object_nerf_amd import checkpoint


@pytest.mark.parametrize("sname", ["sparse", "plain"])
def test_state_dict_round_trip(sname):
    sc = cases.scene_for(A, sname)
    sd = checkpoint.export_state_dict(sc)
    assert "nerf_coarse.xyz_encoding_1.0.weight" in sd and "nerf_fine.inst_rgb.0.bias" in sd
    assert "code_library.embedding_instance.weight" in sd
    if sname != "plain":
        assert sd["embedding_xyz.voxel_idx_map"].dtype == torch.int64
    buf = io.BytesIO()
    torch.save({"state_dict": sd, "epoch": 3}, buf)       # what a Lightning .ckpt holds
    buf.seek(0)
    sc2 = checkpoint.build_from_state_dict(torch.load(buf, weights_only=True))
    sd2 = checkpoint.export_state_dict(sc2)
    assert list(sd) == list(sd2)
    for k in sd:
        assert torch.equal(sd[k], sd2[k]), k
    assert type(sc2.embeddings["xyz"]) is type(sc.embeddings["xyz"])


def test_mismatched_config_is_rejected():
    sd = checkpoint.export_state_dict(cases.scene_for(A, "plain"))
    with pytest.raises(RuntimeError):
        checkpoint.build_from_state_dict(sd, A.default_model_config(use_voxel_embedding=True))


def test_reference_written_checkpoint_fixture_loads_and_the_oracle_reproduces_the_reference_render():
    """tests/golden/reference_small.ckpt (the real train.py system over the reference's module types, oracle/ref_callers.py
    `checkpoint`): strict load into the drop-in types, export equal to the file, and the CPU oracle fed with the LOADED
    parameters is bit-equal to what the reference rendered from that system -- the key map and every tensor arrive where
    they belong (the `-m gpu` twin renders the same file on the device, tests/test_gpu_checkpoint.py)."""
    import os
    import helpers as H
    from oracle import objnerf_oracle as O
    ckpt = torch.load(os.path.join(cases.GOLDEN_DIR, "reference_small.ckpt"), map_location="cpu", weights_only=True)
    sd = ckpt["state_dict"]
    assert sd["embedding_xyz.voxel_idx_map"].dtype == torch.int64 and sd["embedding_xyz.embedding_space_ftr.weight"].shape == (6500, 24)
    sc = checkpoint.build_from_state_dict(ckpt)
    exported = checkpoint.export_state_dict(sc)
    assert list(exported) == list(sd)
    for k in sd:
        assert exported[k].dtype == sd[k].dtype and torch.equal(exported[k], sd[k]), k
    g = cases.load_golden("ckpt_render")
    rays, ids, _, _ = cases.render_inputs("voxel_eval")
    with torch.no_grad():
        codes = sc.code_library.embedding_instance.weight[ids]
        r = O.render_rays(H.state(sc.models["coarse"]), H.state(sc.models["fine"]), H.oracle_grid(sc.embeddings["xyz"]), rays,
                          N_samples=64, N_importance=64, embedding_instance=codes, is_eval=True)
    for k, v in r.items():
        assert torch.equal(v, g["single_" + k]), k
