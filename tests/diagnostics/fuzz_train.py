#!/usr/bin/env python3
"""Wider sweep of tests/test_hip_train.py::test_random_structures_train_step_against_float64_oracle (GPU):
python tests/diagnostics/fuzz_train.py [first] [count].  A case beyond the tolerance is then classified by tests/diagnostics/fuzz_train_masks.py:
"flip" when the HIP step took a discrete decision (a ReLU mask element, the read that wins the final max) differently from
the float64 oracle on a value at the fp32 rounding level (< 1e-5 of the tensor's scale), FAILED otherwise."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "diagnostics"))
import test_hip_train as T      # noqa: E402
from fuzz_train_masks import decision_differences   # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 6
count = int(sys.argv[2]) if len(sys.argv) > 2 else 48
bad = flips = 0
for seed in range(first, first + count):
    try:
        worst, branch = T.run_random_train_case(seed)
        print("seed %d ok, %s branch (worst gradient %s at %.2g of its max)" % (seed, branch, worst[0], worst[1]), flush=True)
    except (AssertionError, RuntimeError, ValueError) as e:
        diffs = [d for d in decision_differences(seed) if d[1]] if isinstance(e, AssertionError) else []
        if diffs and all(rel < 1e-5 for _, _, rel in diffs) and "grad " in str(e):
            flips += 1
            print("seed %d flip (%s): %s" % (seed, "; ".join("%s x%d at %.0e" % d for d in diffs), str(e)[-110:]), flush=True)
        else:
            bad += 1
            print("seed", seed, "FAILED:", str(e)[:600], flush=True)
print("%d structures, %d rounding flips, %d failures" % (count, flips, bad))
sys.exit(1 if bad else 0)
