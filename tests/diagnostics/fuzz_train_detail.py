#!/usr/bin/env python3
"""Per-tensor view of one tests/diagnostics/fuzz_train.py case (GPU): python tests/diagnostics/fuzz_train_detail.py SEED [SEED...]
columns: relative error of the HIP gradient vs the float64 oracle, the fp32 oracle's own relative distance, max |g|."""
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

for seed in [int(a) for a in sys.argv[1:]]:
    kw, cfg, sd, batch, hp, tg, masks = T.random_train_case(seed)
    ohp = O.TrainHyper(**{k: getattr(hp, k) for k in O.TrainHyper.__dataclass_fields__})
    w64 = O.train_step_oracle(sd, cfg, batch.arrays(), tg, ohp, dropout_masks=masks, dtype=torch.float64)
    w32 = O.train_step_oracle(sd, cfg, batch.arrays(), tg, ohp, dropout_masks=masks)
    tr = DanTrainer(cfg, hp, max_batch=8).load_state_dict(sd)
    out = tr.train_step(batch.arrays(), tg, dropout_masks=masks)
    print("seed", seed, kw, "dropout", hp.dropout, "B", len(tg["label"]))
    print("  loss %.6f (oracle %.6f)  norm %.6g (oracle %.6g)" % (out["loss"], float(w64["loss"]), out["grad_norm"], float(w64["grad_norm"])))
    rows = []
    for k, g in w64.items():
        if not k.startswith("grad:"):
            continue
        name = k[5:]
        q = name
        if name.startswith("conv2hidden."):
            idx = sorted({int(x.split(".")[1]) for x in (kk[5:] for kk in w64 if kk.startswith("grad:conv2hidden."))})
            q = "fc.%d.%s" % (idx.index(int(name.split(".")[1])), name.split(".")[2])
        got = tr.tensor("grad:" + q, g.shape)
        sc = max(float(np.abs(g).max()), 1e-30)
        rows.append((float(np.abs(got - g).max()) / sc, float(np.abs(w32[k] - g).max()) / sc, sc, name))
    for e, s32, sc, name in sorted(rows, reverse=True)[:12]:
        print("  %-42s hip %.2e   oracle32 %.2e   max|g| %.3g" % (name, e, s32, sc))
    tr.close()
