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
    sc2 = checkpoint.build_from_state_dict(torch.load(buf))
    sd2 = checkpoint.export_state_dict(sc2)
    assert list(sd) == list(sd2)
    for k in sd:
        assert torch.equal(sd[k], sd2[k]), k
    assert type(sc2.embeddings["xyz"]) is type(sc.embeddings["xyz"])


def test_mismatched_config_is_rejected():
    sd = checkpoint.export_state_dict(cases.scene_for(A, "plain"))
    with pytest.raises(RuntimeError):
        checkpoint.build_from_state_dict(sd, A.default_model_config(use_voxel_embedding=True))
