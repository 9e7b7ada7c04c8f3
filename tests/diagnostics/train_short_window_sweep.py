#!/usr/bin/env python3
"""Which window lengths / row counts the training step's gradients go wrong at (round 5: seeds 20 / 21 of
test_short_windows_train_step_against_float64_oracle).  Same structure, L and the batch shape swept; worst error per tensor group.
Usage (GPU box): python tests/diagnostics/train_short_window_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import test_hip_train as M
from dl4vc_amd.train import DanTrainer
T = M.T


def run(seed, L, R, B):
    kw, cfg, sd, batch, hp, tg, masks = M.random_train_case(seed, length=L, reads=R, sites=B)
    ohp = T.TrainHyper(**{k: getattr(hp, k) for k in T.TrainHyper.__dataclass_fields__})
    want = T.train_step_oracle(sd, cfg, batch.arrays(), tg, ohp, dropout_masks=masks, dtype=torch.float64)
    tr = DanTrainer(cfg, hp, max_batch=8).load_state_dict(sd)
    tr.train_step(batch.arrays(), tg, dropout_masks=masks)
    errs = {}
    for k, g in ((k[5:], v) for k, v in want.items() if k.startswith("grad:")):
        name = k
        if k.startswith("conv2hidden."):
            idx = sorted({int(q[5:].split(".")[1]) for q in want if q.startswith("grad:conv2hidden.")})
            name = "fc.%d.%s" % (idx.index(int(k.split(".")[1])), k.split(".")[2])
        got = tr.tensor("grad:" + name, g.shape).astype(np.float64)
        errs[k] = float(np.abs(got - g).max()) / max(float(np.abs(g).max()), 1e-30)
    tr.close()
    bad = {k: v for k, v in errs.items() if v > 1e-4}
    print("seed %d L %3d R %2d B %d rows %3d: %d tensors above 1e-4%s" % (
        seed, L, R, B, R * B, len(bad), "" if not bad else "; worst " + ", ".join("%s %.1e" % kv for kv in sorted(bad.items(), key=lambda t: -t[1])[:4])), flush=True)


for L in (40, 48, 56, 63, 64, 65, 72, 80, 96, 100, 104, 112, 120, 128):
    run(20, L, 13, 6)
for R, B in ((13, 1), (13, 2), (13, 4), (5, 6), (16, 6), (32, 2), (1, 6)):
    run(20, 64, R, B)
for L in (40, 64, 100, 128, 160, 201):
    run(21, L, 9, 8)
