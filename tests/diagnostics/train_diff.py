#!/usr/bin/env python3
"""GPU debugging aid: per-tensor gradient error of the HIP training step against the float64 training oracle at production
width, for the direct and the Winograd form of the 3-tap layers.  Usage: python tests/diagnostics/train_diff.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dl4vc_amd.config import DanConfig                     # noqa: E402
from dl4vc_amd.train import DanTrainer, TrainHyper, example_weights   # noqa: E402
from dl4vc_amd import synth                                # noqa: E402
from oracle.dan_oracle import random_state_dict            # noqa: E402
from oracle import dan_train_oracle as T                   # noqa: E402

base = dict(reads=12, fc_sizes=(64, 32))
cfg0 = DanConfig(**base)
sd = random_state_dict(cfg0, seed=17)
for k in ("fcHidden2BinTarget", "fcHidden2VT", "fcHidden2AF", "fcHidden2Coverage", "fcHidden2VB", "fcHidden2VR"):
    sd[k + ".weight"] = (sd[k + ".weight"] * np.float32(0.05)).astype(np.float32)
B = 5
batch = synth.make_sites(B, reads=cfg0.reads, seed=18)
rng = np.random.default_rng(19)
hp = TrainHyper()
tg = {"label": np.array([0, 2, 1, 0, 2]), "var_type": np.array([1, 0, 2, 2, 0]), "allele_freq": rng.random(B).astype(np.float32),
      "coverage": rng.integers(5, 60, B).astype(np.float32), "var_base_enum": np.array([1, 2, 5, 8, 3]),
      "var_ref_enum": np.array([4, 3, 1, 2, 2]), "is_snp": np.array([1, 1, 0, 0, 1], np.uint8)}
tg["weight"] = example_weights(tg["is_snp"], hp)
masks = [(rng.random((B, w)) >= hp.dropout).astype(np.uint8) for w in (cfg0.feature_width, 64, 32)]
ohp = T.TrainHyper(**{k: getattr(hp, k) for k in T.TrainHyper.__dataclass_fields__})
w64 = T.train_step_oracle(sd, cfg0, batch.arrays(), tg, ohp, dropout_masks=masks, dtype=torch.float64)
w32 = T.train_step_oracle(sd, cfg0, batch.arrays(), tg, ohp, dropout_masks=masks)
res = {}
for algo in (1, 2):
    tr = DanTrainer(DanConfig(conv_algo=algo, **base), hp, max_batch=8).load_state_dict(sd)
    tr.train_step(batch.arrays(), tg, dropout_masks=masks)
    for k, g in w64.items():
        if not k.startswith("grad:"):
            continue
        name = k[5:]
        if name.startswith("conv2hidden."):
            name = "fc.%d.%s" % ((int(name.split(".")[1]) - 1) // 3, name.split(".")[2])
        got = tr.tensor("grad:" + name, g.shape)
        res.setdefault(k[5:], {})[algo] = float(np.abs(got - g).max()) / max(float(np.abs(g).max()), 1e-30)
    tr.close()
print("%-42s %10s %10s %10s" % ("tensor", "oracle32", "direct", "winograd"))
for k, v in sorted(res.items(), key=lambda kv: -kv[1][2]):
    g = w64["grad:" + k]
    o32 = float(np.abs(w32["grad:" + k] - g).max()) / max(float(np.abs(g).max()), 1e-30)
    print("%-42s %10.2e %10.2e %10.2e" % (k, o32, v[1], v[2]))

# per-layer train-mode activations (x_l) of both forms against the float32 oracle's taps
taps = T.train_step_oracle(sd, cfg0, batch.arrays(), tg, ohp, dropout_masks=masks, taps=True)
R, L = cfg0.reads, cfg0.length
for algo in (1, 2):
    tr = DanTrainer(DanConfig(conv_algo=algo, **base), hp, max_batch=8).load_state_dict(sd)
    tr.backward(batch.arrays(), tg, dropout_masks=masks)
    errs = []
    for l in range(1, cfg0.layers + 1):
        x = tr.debug_buffer("act:x%d" % l, B * R * L * 128).reshape(B, R, L, 128)
        ref = taps["tap:conv%d" % l]
        errs.append(float(np.abs(np.transpose(x, (0, 3, 1, 2)) - ref).max()) / float(np.abs(ref).max()))
    print("algo %d  x_l rel err per layer: %s" % (algo, " ".join("%.1e" % e for e in errs)))
    tr.close()
