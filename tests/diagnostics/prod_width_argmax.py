#!/usr/bin/env python3
"""Where the HIP training step's final max over reads picks another read than the float64 oracle (production width, one seed
of tests/diagnostics/prod_width_seeds.py): the oracle's top-1 minus top-2 at those places -- zero means an exact tie (two reads
with identical receptive fields), a few ulps a rounding matter.   Usage (GPU box): prod_width_argmax.py [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from dl4vc_amd import synth
from dl4vc_amd.config import DanConfig
from oracle.dan_oracle import random_state_dict
from dl4vc_amd.train import DanTrainer, TrainHyper, example_weights
from oracle import dan_train_oracle as T

s = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = DanConfig(reads=12, fc_sizes=(64, 32))
B = 5
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
want = T.train_step_oracle(sd, cfg, batch.arrays(), tg, ohp, dropout_masks=masks, dtype=torch.float64, taps=True)
w32 = T.train_step_oracle(sd, cfg, batch.arrays(), tg, ohp, dropout_masks=masks, taps=True)
tr = DanTrainer(cfg, hp, max_batch=8).load_state_dict(sd)
tr.backward(batch.arrays(), tg, dropout_masks=masks)
R, L = cfg.reads, cfg.length
x = tr.debug_buffer("act:x%d" % cfg.layers, B * R * L * 128).reshape(B, R, L, 128).transpose(0, 3, 1, 2)      # (B, C, R, L)
ref = want["tap:conv%d" % cfg.layers]
r32 = w32["tap:conv%d" % cfg.layers]
print("x7: HIP vs f64 max abs %.3g, torch fp32 vs f64 %.3g (|x| max %.3g)" % (np.abs(x - ref).max(), np.abs(r32 - ref).max(), np.abs(ref).max()))
ah, a64, a32 = x.argmax(2), ref.argmax(2), r32.argmax(2)
srt = np.sort(ref, axis=2)
gap = srt[:, :, -1] - srt[:, :, -2]
for name, a in (("HIP", ah), ("torch fp32", a32)):
    d = np.argwhere(a != a64)
    print("%s: %d of %d max decisions differ from float64; oracle gap at those: exact ties %d, min nonzero %.3g, median %.3g, max %.3g" % (
        name, len(d), a64.size, int((gap[a != a64] == 0).sum()), (gap[a != a64][gap[a != a64] > 0].min() if (gap[a != a64] > 0).any() else 0),
        np.median(gap[a != a64]) if len(d) else 0, gap[a != a64].max() if len(d) else 0))
    for b, c, p in d[:6]:
        print("    site %d ch %d pos %d: f64 picks read %d, %s read %d; f64 values %.9g / %.9g; %s values %.9g / %.9g" % (
            b, c, p, a64[b, c, p], name, a[b, c, p], ref[b, c, a64[b, c, p], p], ref[b, c, a[b, c, p], p], name,
            (x if name == "HIP" else r32)[b, c, a64[b, c, p], p], (x if name == "HIP" else r32)[b, c, a[b, c, p], p]))
print("exact ties in the float64 oracle's max: %d of %d" % (int((gap == 0).sum()), gap.size))
tr.close()
