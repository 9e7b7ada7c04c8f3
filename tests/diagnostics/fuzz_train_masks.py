#!/usr/bin/env python3
"""Why a tests/diagnostics/fuzz_train.py case differs: the discrete decisions of the HIP step (conv / bottleneck / highway ReLU masks, the
read that wins the final max) against the float64 oracle's (GPU): python tests/diagnostics/fuzz_train_masks.py SEED [SEED...].
A decision that differs on a pre-activation (or a max gap) at the fp32 rounding level is a rounding flip -- it moves the
gradients of its layer by a whole element, which no fp32 implementation can avoid; one that differs on a large value is a bug."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch                    # noqa: E402
import test_hip_train as T      # noqa: E402
from oracle import dan_train_oracle as O   # noqa: E402
from dl4vc_amd.train import DanTrainer     # noqa: E402


def decision_differences(seed, log=None):
    """[(where, count, largest |oracle value| among the differing decisions relative to the tensor's scale)]"""
    kw, cfg, sd, batch, hp, tg, masks = T.random_train_case(seed)
    ohp = O.TrainHyper(**{k: getattr(hp, k) for k in O.TrainHyper.__dataclass_fields__})
    w64 = O.train_step_oracle(sd, cfg, batch.arrays(), tg, ohp, dropout_masks=masks, dtype=torch.float64, taps=True)
    tr = DanTrainer(cfg, hp, max_batch=8).load_state_dict(sd)
    tr.backward(batch.arrays(), tg, dropout_masks=masks)
    B, R, L = len(tg["label"]), cfg.reads, cfg.length
    out = []

    def rows(name, width, C):                                       # [B][R][L][width] device layout -> (B, C, R, L)
        return np.transpose(tr.debug_buffer(name, B * R * L * width).reshape(B, R, L, width)[..., :C], (0, 3, 1, 2))

    for l in range(1, cfg.layers + 1):
        pre = w64["tap:pre%d" % l]
        d = (rows("act:a%d" % l, 128, pre.shape[1]) > 0) != (pre > 0)
        out.append(("conv %d ReLU" % l, int(d.sum()), float(np.abs(pre[d]).max() / np.abs(pre).max()) if d.any() else 0.0))
        if cfg.bottleneck > 0:
            hpre = w64["tap:hpre%d" % l]
            d = (rows("act:h%d" % l, 32, hpre.shape[1]) > 0) != (hpre > 0)
            out.append(("bottleneck %d ReLU" % l, int(d.sum()), float(np.abs(hpre[d]).max() / np.abs(hpre).max()) if d.any() else 0.0))
    xo = w64["tap:conv%d" % cfg.layers]
    xl = rows("act:x%d" % cfg.layers, 128, xo.shape[1])
    am, ao = xl.argmax(axis=2), xo.argmax(axis=2)
    d = am != ao
    gap = np.take_along_axis(xo, ao[:, :, None, :], 2)[:, :, 0, :] - np.take_along_axis(xo, am[:, :, None, :], 2)[:, :, 0, :]
    # (exact ties -- identical reads -- are not decisions: whichever read takes the gradient, every parameter gradient is the same)
    real = d & (gap > 1e-12 * np.abs(xo).max())
    out.append(("final max over reads", int(real.sum()), float(gap[real].max() / np.abs(xo).max()) if real.any() else 0.0))
    if cfg.bottleneck > 0:
        F = tr.query("feature_width")
        feat = tr.debug_buffer("feature", B * tr.query("feature_stride")).reshape(B, -1)[:, :F]
        fo = w64["tap:feature"]
        hw0 = 2 * xo.shape[1] * L
        d = (feat[:, hw0:] > 0) != (fo[:, hw0:] > 0)
        out.append(("highway ReLU", int(d.sum()), 0.0))
    tr.close()
    if log:
        log("seed %d %s" % (seed, kw))
        for where, n, rel in out:
            if n:
                log("  %s: %d decisions differ (largest oracle value among them %.1e of the tensor's scale)" % (where, n, rel))
    return out


if __name__ == "__main__":
    for seed in [int(a) for a in sys.argv[1:]]:
        decision_differences(seed, print)
