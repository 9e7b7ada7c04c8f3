"""bench.py reports profiled HBM traffic (`roofline.traffic`) only for the code it was measured on.

profiles/rNN_traffic.json carries, per section, the `source_hash` of the library the PMC passes ran on (dan_source_hash():
sha256 over the kernel + C-ABI sources, compiled in by csrc/Makefile) and the `chunk_sites` its handle chose; bench.py compares
both with the library it has loaded and the handle it times.  CPU only: no compute call is made."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from dl4vc_amd import capi  # noqa: E402


def write(tmp_path, name, rec):
    with open(os.path.join(tmp_path, name), "w") as f:
        json.dump(rec, f)


def test_traffic_is_reported_only_for_the_same_sources_and_chunk(tmp_path):
    h = "0123456789abcdef"
    write(tmp_path, "r07_traffic.json", {"segment_kernel_bytes_per_launch": {"total": 36.3e9, "source_hash": h, "chunk_sites": 2048},
                                          "train_step_b64": {"hbm_bytes_per_step": 87.3e9, "source_hash": h}})
    d = str(tmp_path)
    assert bench.pmc_traffic("segment_kernel_bytes_per_launch", "total", h, 2048, d) == (36300000000, False)
    # a doctored hash (= the kernels changed since the capture): null + stale
    assert bench.pmc_traffic("segment_kernel_bytes_per_launch", "total", "fedcba9876543210", 2048, d) == (None, True)
    # same sources, but the handle sized its chunk differently (less free device memory): the bytes per launch are not these
    assert bench.pmc_traffic("segment_kernel_bytes_per_launch", "total", h, 1024, d) == (None, True)
    # a section without a chunk (the training step) is tied by the hash alone
    assert bench.pmc_traffic("train_step_b64", "hbm_bytes_per_step", h, None, d) == (87300000000, False)
    assert bench.pmc_traffic("train_step_b64", "hbm_bytes_per_step", "unknown", None, d) == (None, True)
    # nothing captured for a section: absent, not stale
    assert bench.pmc_traffic("train_step_b10", "hbm_bytes_per_step", h, None, d) == (None, False)


def test_a_capture_without_identity_is_stale(tmp_path):
    """Rounds 1-4 wrote no hash: such a file can never vouch for the current kernels."""
    write(tmp_path, "r04_traffic.json", {"segment_kernel_bytes_per_launch": {"total": 36.3e9}})
    assert bench.pmc_traffic("segment_kernel_bytes_per_launch", "total", capi.tree_source_hash(), 2048, str(tmp_path)) == (None, True)


def test_the_newest_round_wins(tmp_path):
    h = "0123456789abcdef"
    write(tmp_path, "r04_traffic.json", {"s": {"total": 1, "source_hash": h}})
    write(tmp_path, "r05_traffic.json", {"s": {"total": 2, "source_hash": h}})
    assert bench.pmc_traffic("s", "total", h, None, str(tmp_path)) == (2, False)


def test_no_profiles_at_all(tmp_path):
    assert bench.pmc_traffic("s", "total", "x", None, str(tmp_path)) == (None, False)


def test_library_hash_equals_the_tree_it_was_built_from():
    """The library in the tree is a build of the sources in the tree (what __graft_entry__.build() leaves behind), and the
    Makefile's digest and the Python one are the same computation."""
    if not os.path.isfile(capi.LIB_PATH):
        pytest.skip("library not built")
    assert capi.source_hash() == capi.tree_source_hash()
    assert len(capi.source_hash()) == 16 and capi.source_hash() != "unknown"


def test_one_changed_source_byte_changes_the_hash(tmp_path):
    src = os.path.join(ROOT, "dl4vc_amd", "csrc")
    for f in os.listdir(src):
        if f.endswith((".hip", ".h", ".cpp")):
            shutil.copy(os.path.join(src, f), tmp_path)
    assert capi.tree_source_hash(str(tmp_path)) == capi.tree_source_hash()
    with open(os.path.join(tmp_path, "dan_kernels.hip"), "a") as f:
        f.write("\n")
    assert capi.tree_source_hash(str(tmp_path)) != capi.tree_source_hash()
    # sources outside the hashed set (the host-only loader) do not take part
    shutil.copy(os.path.join(src, "dan_kernels.hip"), tmp_path)
    with open(os.path.join(tmp_path, "dan_loader.cpp"), "a") as f:
        f.write("\n")
    assert capi.tree_source_hash(str(tmp_path)) == capi.tree_source_hash()
    # ... and the Makefile computes the same digest
    out = subprocess.check_output("cat $(ls dan_*.hip dan_*.h dan_capi.cpp dan_train_capi.cpp | LC_ALL=C sort) | sha256sum | cut -c1-16",
                                  shell=True, cwd=src, text=True).strip()
    assert out == capi.tree_source_hash()
