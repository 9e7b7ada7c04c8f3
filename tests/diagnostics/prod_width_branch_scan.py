#!/usr/bin/env python3
"""Which (sites, reads, pileup seed) combinations of the production-width training check take the TIGHT branch (no ReLU / max
decision differs from the float64 oracle's) on the current kernels: the committed cases of
tests/test_hip_train.py::test_production_width_step_against_oracle are chosen from this scan so that both branches are exercised.
Usage (GPU box): python tests/diagnostics/prod_width_branch_scan.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_train as M
from dl4vc_amd.config import DanConfig

for B, R in ((2, 4), (1, 8), (3, 6)):
    for seed in range(30, 36):
        try:
            rec = M._full_width_step_check(DanConfig(reads=R, fc_sizes=(64, 32)), B, "scan B %d R %d seed %d" % (B, R, seed), None,
                                           data_seed=seed, report="scan_B%d_R%d_seed%d" % (B, R, seed))
            print("B %d R %d seed %d: %s, worst %s" % (B, R, seed, rec["branch"], rec["worst_gradient"]), flush=True)
        except AssertionError as e:
            print("B %d R %d seed %d: ASSERTION %s" % (B, R, seed, str(e)[:300]), flush=True)
