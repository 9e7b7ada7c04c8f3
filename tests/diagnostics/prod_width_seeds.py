#!/usr/bin/env python3
"""How far one fp32 training step at production width sits from the FLOAT64 oracle, over several seeds: the quantity that
tests/test_hip_train.py::test_production_width_step_against_oracle bounds by 1e-4 + 2 x (the fp32 oracle's own distance).
Prints, per seed, the worst tensor's error in units of that bound and in units of the fp32 oracle's distance -- the spread says
whether the bound is a property of the implementation or of fp32 rounding noise (ReLU / max-pool decisions flipping).

Usage (GPU box): python tests/diagnostics/prod_width_seeds.py [n_seeds]      (DL4VC_DAN_LIB selects another build of the library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from dl4vc_amd import synth
from dl4vc_amd.config import DanConfig
from oracle.dan_oracle import random_state_dict
from dl4vc_amd.train import DanTrainer, TrainHyper, example_weights
from oracle import dan_train_oracle as T

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = DanConfig(reads=12, fc_sizes=(64, 32))
B = 5
rows = []
for s in range(n_seeds):
    sd = random_state_dict(cfg, seed=17 + 100 * s)
    for k in ("fcHidden2BinTarget", "fcHidden2VT", "fcHidden2AF", "fcHidden2Coverage", "fcHidden2VB", "fcHidden2VR"):
        sd[k + ".weight"] = (sd[k + ".weight"] * np.float32(0.05)).astype(np.float32)
    batch = synth.make_sites(B, reads=cfg.reads, seed=18 + 100 * s)
    rng = np.random.default_rng(19 + 100 * s)
    hp = TrainHyper()
    tg = {"label": rng.integers(0, 3, B), "var_type": rng.integers(0, 3, B), "allele_freq": rng.random(B).astype(np.float32),
          "coverage": rng.integers(5, 60, B).astype(np.float32), "var_base_enum": rng.integers(1, 9, B),
          "var_ref_enum": rng.integers(1, 5, B), "is_snp": rng.integers(0, 2, B).astype(np.uint8)}
    tg["weight"] = example_weights(tg["is_snp"], hp)
    masks = [(rng.random((B, w)) >= hp.dropout).astype(np.uint8) for w in (cfg.feature_width, 64, 32)]
    ohp = T.TrainHyper(**{k: getattr(hp, k) for k in T.TrainHyper.__dataclass_fields__})
    want = T.train_step_oracle(sd, cfg, batch.arrays(), tg, ohp, dropout_masks=masks, dtype=torch.float64)
    w32 = T.train_step_oracle(sd, cfg, batch.arrays(), tg, ohp, dropout_masks=masks)
    tr = DanTrainer(cfg, hp, max_batch=8).load_state_dict(sd)
    tr.train_step(batch.arrays(), tg, dropout_masks=masks)
    worst = (0.0, "", 0.0, 0.0)
    outside = []
    for k, g in ((k[5:], v) for k, v in want.items() if k.startswith("grad:")):
        name = k
        if k.startswith("conv2hidden."):
            idx = sorted({int(q[5:].split(".")[1]) for q in want if q.startswith("grad:conv2hidden.")})
            name = "fc.%d.%s" % (idx.index(int(k.split(".")[1])), k.split(".")[2])
        got = tr.tensor("grad:" + name, g.shape).astype(np.float64)
        scale = max(float(np.abs(g).max()), 1e-30)
        err = float(np.abs(got - g).max()) / scale
        slack = float(np.abs(w32["grad:" + k] - g).max()) / scale
        ratio = err / (1e-4 + 2 * slack)
        n_out = int((np.abs(got - g) > (1e-4 + 2 * slack) * scale).sum())
        outside.append((k, n_out, g.size))
        if ratio > worst[0]:
            worst = (ratio, k, err, slack)
    tr.close()
    rows.append(worst)
    print("seed %d: worst %.2f of the bound (%s: HIP %.2e of max, fp32 oracle %.2e of max)" % (s, *worst), flush=True)
    bad = [(k, n, m) for k, n, m in outside if n]
    print("        tensors with elements outside the bound: %d of %d; elements outside / size: %s" % (
        len(bad), len(outside), ", ".join("%s %d/%d" % (k.replace("conv1D_", "").replace("_layers", ""), n, m) for k, n, m in sorted(bad, key=lambda t: -t[1] / t[2])[:6])), flush=True)
print("max over seeds: %.2f of the bound; seeds above it: %d / %d" % (max(r[0] for r in rows), sum(r[0] > 1 for r in rows), n_seeds))
