"""Drop-in `CodeLibrary` (reference: models/code_library.py:5-28): per-object latent codes.

A 64x64 table lookup is index plumbing, not arithmetic: it stays a torch gather on the device
(the renderer reads the resulting (N,64) rows, or the table row itself for a constant id).
"""
import torch
from torch import nn


class CodeLibrary(nn.Module):
    def __init__(self, model_config):
        super().__init__()
        self.embedding_instance = nn.Embedding(
            model_config.get("N_max_objs", 64), model_config.get("N_obj_code_length", 64))

    def forward(self, inputs):
        ret = {}
        if "instance_ids" in inputs:
            ret["embedding_instance"] = self.embedding_instance(inputs["instance_ids"].squeeze())
        return ret
