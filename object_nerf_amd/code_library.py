"""Per-object latent codes -- drop-in for the reference's `CodeLibrary` (models/code_library.py:5-28).

Contract kept for callers and checkpoints: constructor takes the model config (`.get` access), the table is
an `nn.Embedding` registered as `embedding_instance` (state_dict key `embedding_instance.weight`; read directly
by render_tools/multi_rendering.py:46 and tools/extract_mesh.py:99), and calling the module with a batch dict
returns `{"embedding_instance": rows}` when the batch carries `instance_ids`, `{}` otherwise.

A 64 x 64 table lookup is index plumbing, not arithmetic: it stays a torch gather on the device (which also gives
the code table its gradient through ordinary autograd on the training path).
"""
from torch import nn


class CodeLibrary(nn.Module):
    """A boundary type: three statements the callers and the checkpoint layout dictate (attribute name, `.get` defaults,
    the squeeze, the result key), so it necessarily reads like the reference's."""

    def __init__(self, model_config):
        super().__init__()
        self.embedding_instance = nn.Embedding(model_config.get("N_max_objs", 64), model_config.get("N_obj_code_length", 64))

    def forward(self, inputs):
        if "instance_ids" not in inputs:
            return {}
        return {"embedding_instance": self.embedding_instance(inputs["instance_ids"].squeeze())}
