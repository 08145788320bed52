#!/usr/bin/env python
"""Bit digests of every synthetic input the goldens depend on, as generated on THIS host, and the differences from
tests/golden/input_digests.json (the host that made the goldens).  usage: python tools/input_digests.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402

if __name__ == "__main__":
    here = cases.input_digests()
    path = os.path.join(cases.GOLDEN_DIR, "input_digests.json")
    gold = json.load(open(path)) if os.path.exists(path) else {}
    diff = sorted(k for k in set(here) | set(gold) if here.get(k) != gold.get(k))
    print(json.dumps({"cpu_capability": torch.backends.cpu.get_cpu_capability(), "n": len(here), "differ_from_golden_host": diff}, indent=1))
