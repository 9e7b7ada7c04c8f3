#!/usr/bin/env python3
"""One-off wider sweep of tests/test_hip_long_window.py::test_long_window_random_structures_all_forms_agree (GPU):
    python tests/diagnostics/fuzz_long_windows.py [first] [count]
Random structures at 209..304 columns (dilations drawn too) through fp32 Winograd / direct / skip / chunks and bf16x3."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest                      # noqa: E402
import test_hip_long_window as T   # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 110
count = int(sys.argv[2]) if len(sys.argv) > 2 else 48
bad = skipped = 0
for seed in range(first, first + count):
    try:
        T.test_long_window_random_structures_all_forms_agree(seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, T._random_long_structure(seed)[0], "FAILED:", str(e)[:300], flush=True)
    except pytest.skip.Exception as e:
        skipped += 1
        print("seed", seed, "skipped:", str(e)[:100], flush=True)
print("%d structures, %d refused by dan_create, %d failures" % (count, skipped, bad))
sys.exit(1 if bad else 0)
