#!/usr/bin/env python3
"""Are the sporadic gradient misses of the random training structures ReLU / max decisions on a rounding edge (as at production
width) or something else?  For each failing (seed, L) of tests/diagnostics/train_short_window_sweep.py: the decisions of the HIP
step that differ from the float64 oracle's, with the oracle's operand at the decision.
Usage (GPU box): python tests/diagnostics/train_flip_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import test_hip_train as M
from dl4vc_amd.train import DanTrainer
T = M.T

for seed, L, R, B in [(20, 64, 13, 6), (21, 40, 9, 8), (21, 201, 9, 8), (21, 64, 9, 8), (20, 65, 13, 6)]:
    kw, cfg, sd, batch, hp, tg, masks = M.random_train_case(seed, length=L, reads=R, sites=B)
    ohp = T.TrainHyper(**{k: getattr(hp, k) for k in T.TrainHyper.__dataclass_fields__})
    want = T.train_step_oracle(sd, cfg, batch.arrays(), tg, ohp, dropout_masks=masks, dtype=torch.float64, taps=True)
    w32 = T.train_step_oracle(sd, cfg, batch.arrays(), tg, ohp, dropout_masks=masks, taps=True)
    tr = DanTrainer(cfg, hp, max_batch=8).load_state_dict(sd)
    tr.train_step(batch.arrays(), tg, dropout_masks=masks)
    diff = [d for d in M.decisions_differing(tr, want, cfg, B) if d[2]]
    # the fp32 torch oracle's own flips against float64
    o32 = []
    for l in range(1, cfg.layers + 1):
        a, b = w32["tap:pre%d" % l], want["tap:pre%d" % l]
        d = (a > 0) != (b > 0)
        if d.any():
            o32.append(("conv", l, int(d.sum()), float(np.abs(b[d]).max())))
    worst = 0.0
    for k, g in ((k[5:], v) for k, v in want.items() if k.startswith("grad:")):
        name = k
        if k.startswith("conv2hidden."):
            idx = sorted({int(q[5:].split(".")[1]) for q in want if q.startswith("grad:conv2hidden.")})
            name = "fc.%d.%s" % (idx.index(int(k.split(".")[1])), k.split(".")[2])
        got = tr.tensor("grad:" + name, g.shape).astype(np.float64)
        worst = max(worst, float(np.abs(got - g).max()) / max(float(np.abs(g).max()), 1e-30))
    tr.close()
    print("seed %d L %d: worst gradient %.2e of max; HIP decisions differing from float64: %s; fp32 torch oracle's: %s" % (seed, L, worst, diff, o32), flush=True)
