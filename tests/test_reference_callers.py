"""CPU, build container only (skipped where /root/reference is absent): the reference's own CALLERS of the hot path --
train.py::ObjectNeRFSystem (__init__, forward chunk loop 73-105, validation_step, training_step + backward) and
render_tools/editable_renderer.py::EditableRenderer (load_model -> Lightning load_from_checkpoint, render_edit,
render_origin) -- executed UNMODIFIED, once over the reference's models/* and once over `dropin/` (SURVEY.md §8 rows
a15, b, f3).  oracle/ref_callers.py explains the harness; each flavour needs its own process (`models`, `train`, ...)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("OBJNERF_REFERENCE_ROOT", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "train.py")), reason="reference tree not mounted")


def run(flavour, work):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_callers.py"), flavour, str(work)],
                       capture_output=True, text=True, cwd=str(work), timeout=600)
    assert r.returncode == 0, "%s flavour failed:\n%s" % (flavour, r.stderr[-3000:])
    return r.stdout.strip().splitlines()[-1]


def same_npz(a, b):
    za, zb = np.load(a), np.load(b)
    assert sorted(za.files) == sorted(zb.files)
    for k in za.files:
        assert za[k].dtype == zb[k].dtype and za[k].shape == zb[k].shape and np.array_equal(za[k], zb[k]), k


def test_reference_callers_run_unchanged_on_the_drop_in(tmp_path):
    # 1. the callers over the reference's own modules: outputs + a Lightning-keyed checkpoint of the reference's types
    assert run("reference", tmp_path).startswith("reference flavour")
    # the committed fixture the GPU replay test grades against IS this run's output
    same_npz(tmp_path / "callers.npz", os.path.join(GOLD, "callers_outputs.npz"))
    # 2. the same files over dropin/: construction by train.py::__init__, strict load of the REFERENCE's checkpoint through
    #    EditableRenderer.load_model, every call bound against the product signatures, outputs bit-equal (asserted inside)
    assert "equal to the reference flavour's" in run("dropin", tmp_path)
    same_npz(tmp_path / "calls.npz", os.path.join(GOLD, "callers_calls.npz"))
    # 3. f3, export direction: the product's exported checkpoint loads strictly into the reference's own module types
    assert run("reference-load", tmp_path).startswith("reference-load")
