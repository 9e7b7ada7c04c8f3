#!/usr/bin/env python3
"""One-off wider sweep of tests/test_hip_parity.py::test_random_structures_all_forms_agree (GPU): python tests/diagnostics/fuzz_structures.py [first] [count]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_parity as T      # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 8
count = int(sys.argv[2]) if len(sys.argv) > 2 else 64
bad = 0
for seed in range(first, first + count):
    try:
        T.test_random_structures_all_forms_agree(seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED:", str(e)[:300])
print("%d structures, %d failures" % (count, bad))
sys.exit(1 if bad else 0)
