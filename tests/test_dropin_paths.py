"""CPU: with `dropin/` first on sys.path the reference's import lines resolve to the MI355X types
(run in a subprocess so the `models` name does not leak into the other tests)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

CODE = r"""
import sys, os
root, ref = sys.argv[1], sys.argv[2]
if os.path.isdir(ref):
    sys.path.insert(0, ref)
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, 'dropin'))
from models.nerf_model import ObjectNeRF
from models.embedding_helper import EmbeddingVoxel, Embedding
from models.rendering import render_rays
from models.code_library import CodeLibrary
from render_tools.multi_rendering import render_rays_multi
for o in (ObjectNeRF, EmbeddingVoxel, Embedding, render_rays, CodeLibrary, render_rays_multi):
    assert o.__module__.startswith('object_nerf_amd.'), o
if os.path.isdir(ref):
    import importlib.util
    import models.losses                       # still the reference's file
    assert models.losses.__file__.startswith(ref)
    assert importlib.util.find_spec('render_tools.editable_renderer').origin.startswith(ref)
print('ok')
"""


def test_reference_import_lines_resolve_to_drop_in():
    r = subprocess.run([sys.executable, "-c", CODE, ROOT, REF], capture_output=True, text=True, cwd="/tmp")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr
