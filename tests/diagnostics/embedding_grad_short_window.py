#!/usr/bin/env python3
"""(HISTORICAL, round 5: the debug tap "du0" this script reads was removed when layer 1's backward went to binned sums -- the training fixtures and
the float64-oracle tests pin the embedding gradient now; kept for the record of what was checked.)
Where the embedding gradient of a short-window training step goes wrong: the step's own du_0 (gradient of the encoded input,
tap "du0") is reduced on the host in float64 with the embedding rules (padding_idx 0, scale_grad_by_freq per lookup) and compared
with (a) the device's grad:embeddings.weight -- isolates the three embedding kernels -- and (b) the float64 oracle's.
Usage (GPU box): python tests/diagnostics/embedding_grad_short_window.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import test_hip_train as M
from dl4vc_amd.train import DanTrainer
T = M.T

for seed, L, R, B in [(20, 64, 13, 6), (21, 40, 9, 8), (22, 100, 16, 7)]:
    kw, cfg, sd, batch, hp, tg, masks = M.random_train_case(seed, length=L, reads=R, sites=B)
    ohp = T.TrainHyper(**{k: getattr(hp, k) for k in T.TrainHyper.__dataclass_fields__})
    want = T.train_step_oracle(sd, cfg, batch.arrays(), tg, ohp, dropout_masks=masks, dtype=torch.float64)
    tr = DanTrainer(cfg, hp, max_batch=8).load_state_dict(sd)
    tr.train_step(batch.arrays(), tg, dropout_masks=masks)
    g_dev = tr.tensor("grad:embeddings.weight", (10, 20)).astype(np.float64)
    du = tr.debug_buffer("du0", B * R * L * 128).reshape(B, R, L, 128).astype(np.float64)
    reads, ref = batch.reads.astype(int), batch.ref.astype(int)
    g_host = np.zeros((10, 20))
    for k in range(1, 10):
        m = reads == k
        if m.any():
            g_host[k] += du[..., 0:20][m].sum(0) / m.sum()
        mr = ref == k
        if mr.any():
            g_host[k] += du[..., 20:40].sum(1)[mr].sum(0) / mr.sum()
    g_or = want["grad:embeddings.weight"]
    sc = np.abs(g_or).max()
    print("seed %d L %d: |dev - host(du0)| %.3g  |host(du0) - oracle| %.3g  |dev - oracle| %.3g   (of max |g| = %.3g)" % (
        seed, L, np.abs(g_dev - g_host).max() / sc, np.abs(g_host - g_or).max() / sc, np.abs(g_dev - g_or).max() / sc, sc))
    per_tok = np.abs(g_dev - g_or).max(1) / sc
    print("   per token |dev - oracle| / max: " + " ".join("%.1e" % v for v in per_tok))
    print("   token counts (reads lookup): " + " ".join(str(int((reads == k).sum())) for k in range(10)))
    print("   token counts (ref lookup):   " + " ".join(str(int((ref == k).sum())) for k in range(10)))
    print("   du0 beyond channel 48 all zero: %s; max |du0| %.3g" % (not np.any(du[..., 48:]), np.abs(du).max()))
    tr.close()

# every gradient tensor of the same cases, not only the first one that fails
print()
for seed, L, R, B in [(20, 64, 13, 6), (21, 40, 9, 8)]:
    kw, cfg, sd, batch, hp, tg, masks = M.random_train_case(seed, length=L, reads=R, sites=B)
    ohp = T.TrainHyper(**{k: getattr(hp, k) for k in T.TrainHyper.__dataclass_fields__})
    want = T.train_step_oracle(sd, cfg, batch.arrays(), tg, ohp, dropout_masks=masks, dtype=torch.float64)
    tr = DanTrainer(cfg, hp, max_batch=8).load_state_dict(sd)
    tr.train_step(batch.arrays(), tg, dropout_masks=masks)
    print("seed %d %s" % (seed, kw))
    for k, g in ((k[5:], v) for k, v in want.items() if k.startswith("grad:")):
        name = k
        if k.startswith("conv2hidden."):
            idx = sorted({int(q[5:].split(".")[1]) for q in want if q.startswith("grad:conv2hidden.")})
            name = "fc.%d.%s" % (idx.index(int(k.split(".")[1])), k.split(".")[2])
        got = tr.tensor("grad:" + name, g.shape).astype(np.float64)
        sc = max(float(np.abs(g).max()), 1e-30)
        print("   %-40s err/max %.2e   max %.3g" % (k, float(np.abs(got - g).max()) / sc, sc))
    tr.close()
