#!/usr/bin/env python
"""per-ray distance of the training_step replay (tests/test_gpu_callers.py) from the reference's outputs, for the two autograd forms"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, helpers as H
import object_nerf_amd as A
import test_gpu_callers as T
DEV = "cuda"
scene = cases.scene_for(A, "voxel", device=DEV)
gold = cases.load_golden("callers_outputs")
calls = [c for c in T.load_calls() if c["scenario"] == "training_step"]
table = scene.code_library.embedding_instance.weight
for p in [p for m in (scene.models["coarse"], scene.models["fine"], scene.code_library, scene.embeddings["xyz"]) for p in m.parameters()]:
    p.requires_grad_(True)
for nodes in ("1", "2"):
    os.environ["OBJNERF_TRAIN_NODES"] = nodes
    chunks = []
    for c in calls:
        rows = c["tens"]["embedding_instance"].to(DEV)
        ids = (rows[:, None, :] == table.detach()[None]).all(-1).float().argmax(1)
        chunks.append(T.issue_render_rays(scene, c, codes=scene.code_library({"instance_ids": ids})["embedding_instance"]))
    out = {k: torch.cat([c[k] for c in chunks], 0).detach() for k in chunks[0]}
    with torch.no_grad():
        inf = [T.issue_render_rays(scene, c) for c in calls]
    outi = {k: torch.cat([c[k] for c in inf], 0) for k in inf[0]}
    print("NODES", nodes)
    for k in sorted(out):
        g = gold["train_" + k]
        d = (out[k].cpu().double() - g.double()).abs().reshape(g.shape[0], -1).max(1).values / g.abs().max().clamp_min(1e-30)
        di = (outi[k].cpu().double() - g.double()).abs().reshape(g.shape[0], -1).max(1).values / g.abs().max().clamp_min(1e-30)
        print("  %-26s train path: max %.2e at ray %d (rays > 1e-3: %s) | inference path: max %.2e" % (k, d.max().item(), int(d.argmax()), [int(i) for i in torch.nonzero(d > 1e-3).flatten()][:8], di.max().item()))
