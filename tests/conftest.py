import os
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
    # The CPU oracle is hundreds of small ATen ops per call: on the GPU box's 256 host cores torch's default (one thread per core)
    # is several times SLOWER than 16 threads (bench.py's cpu_baseline probes this: 16 is its best).  The oracle-heavy GPU tests
    # (gradients against autograd through the oracle, the Adam trajectories) spend most of their time there.
    import torch
    torch.set_num_threads(min(torch.get_num_threads(), 16))
