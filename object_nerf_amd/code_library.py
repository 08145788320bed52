"""Per-object latent codes -- drop-in for the reference's `CodeLibrary` (models/code_library.py:5-28).

Contract kept for callers and checkpoints: constructor takes the model config (`.get` access), the table is
an `nn.Embedding` registered as `embedding_instance` (state_dict key `embedding_instance.weight`; read directly
by render_tools/multi_rendering.py:46 and tools/extract_mesh.py:99), and calling the module with a batch dict
returns `{"embedding_instance": rows}` when the batch carries `instance_ids`, `{}` otherwise.

A 64 x 64 table lookup is index plumbing, not arithmetic: it stays a torch gather on the device (which also gives
the code table its gradient through ordinary autograd on the training path).
"""
from torch import nn

_IDS, _ROWS = "instance_ids", "embedding_instance"


class CodeLibrary(nn.Module):
    def __init__(self, model_config):
        super().__init__()
        table = nn.Embedding(num_embeddings=model_config.get("N_max_objs", 64),
                             embedding_dim=model_config.get("N_obj_code_length", 64))
        self.add_module(_ROWS, table)

    def codes_for(self, instance_ids):
        """(N,1) / (N,) integer ids -> (N, code_length) rows (ids are squeezed like the reference does)"""
        return getattr(self, _ROWS)(instance_ids.squeeze())

    def forward(self, inputs):
        return {_ROWS: self.codes_for(inputs[_IDS])} if _IDS in inputs else {}
