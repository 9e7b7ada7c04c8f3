#!/usr/bin/env python3
"""Per-tensor view of tests/test_hip_train.py::test_production_width_step_strict_bar... (GPU): every gradient tensor's distance
from the float64 oracle once all decisions are off their edges, for the HIP step and for the fp32 torch oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_hip_train as H
from dl4vc_amd.config import DanConfig
from dl4vc_amd import synth
from dl4vc_amd.train import DanTrainer, TrainHyper, example_weights
from oracle.dan_oracle import random_state_dict
from oracle import dan_train_oracle as T

cfg = DanConfig(reads=6, fc_sizes=(64, 32))
sd = random_state_dict(cfg, seed=23)
for k in ("fcHidden2BinTarget", "fcHidden2VT", "fcHidden2AF", "fcHidden2Coverage", "fcHidden2VB", "fcHidden2VR"):
    sd[k + ".weight"] = (sd[k + ".weight"] * np.float32(0.05)).astype(np.float32)
B = 3
batch = synth.make_sites(B, reads=cfg.reads, seed=24)
planes = [a.copy() for a in batch.arrays()]
# every read random and every row non-empty: two reads that follow the reference are nearly identical over long stretches, and
# their outputs -- rivals in the final max -- then differ by less than any margin at hundreds of places
_r = np.random.default_rng(26)
planes[0] = _r.integers(1, 9, planes[0].shape).astype(np.uint8)
planes[1] = _r.integers(2, 42, planes[1].shape).astype(np.uint8)
planes[2] = _r.integers(1, 3, planes[2].shape).astype(np.uint8)
rng = np.random.default_rng(25)
hp = TrainHyper()
tg = {"label": np.array([0, 2, 1]), "var_type": np.array([1, 0, 2]), "allele_freq": rng.random(B).astype(np.float32),
      "coverage": rng.integers(5, 60, B).astype(np.float32), "var_base_enum": np.array([1, 2, 5]),
      "var_ref_enum": np.array([4, 3, 1]), "is_snp": np.array([1, 1, 0], np.uint8)}
tg["weight"] = example_weights(tg["is_snp"], hp)
masks = [(rng.random((B, w)) >= hp.dropout).astype(np.uint8) for w in (cfg.feature_width, 64, 32)]
ohp = T.TrainHyper(**{k: getattr(hp, k) for k in T.TrainHyper.__dataclass_fields__})
fc = sorted(k[:-7] for k in sd if k.startswith("conv2hidden.") and k.endswith(".weight"))
want, it = H._move_decisions_off_their_edges(sd, cfg, planes, tg, ohp, masks)
print("rounds", it)
w32 = T.train_step_oracle(sd, cfg, planes, tg, ohp, dropout_masks=masks)
tr = DanTrainer(cfg, hp, max_batch=4).load_state_dict(sd)
tr.train_step(planes, tg, dropout_masks=masks)
rows = []
for k, g in want.items():
    if not k.startswith("grad:"):
        continue
    name = k[5:]
    our = name
    if name.startswith("conv2hidden."):
        our = "fc.%d.%s" % (fc.index(name.rsplit(".", 1)[0]), name.rsplit(".", 1)[1])
    got = tr.tensor("grad:" + our, g.shape)
    sc = max(float(np.abs(g).max()), 1e-30)
    rows.append((float(np.abs(got - g).max()) / sc, float(np.abs(w32[k] - g).max()) / sc, sc, name))
for r in sorted(rows, reverse=True)[:12]:
    print("%-40s HIP %.2e   torch fp32 %.2e   (max |g| %.3g)" % (r[3], r[0], r[1], r[2]))
R_, L_ = cfg.reads, cfg.length
for l in range(1, cfg.layers + 1):
    a = tr.debug_buffer("act:a%d" % l, B * R_ * L_ * 128).reshape(B, R_, L_, 128).transpose(0, 3, 1, 2)
    pre = want["tap:pre%d" % l]
    flips = int(((a > 0) != (pre > 0)).sum())
    err = float(np.abs(a - np.maximum(pre, 0)).max())
    x = tr.debug_buffer("act:x%d" % l, B * R_ * L_ * 128).reshape(B, R_, L_, 128).transpose(0, 3, 1, 2)
    ex = float(np.abs(x - want["tap:conv%d" % l]).max())
    print("layer %d: relu output max |err| %.2e (max %.3g), mask flips %d, min |pre| %.2e; layer output max |err| %.2e"
          % (l, err, np.abs(pre).max(), flips, np.abs(pre).min(), ex))
