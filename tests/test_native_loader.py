"""Native batched loader (row N1) against the Python host logic and the reference-generated fixtures."""
import ctypes as C
import json
import os
import time

import numpy as np
import pytest

from conftest import GOLDEN
from dl4vc_amd import synth, hdf5io, loader
from dl4vc_amd.alleles import safe_allele_mask_vectors
from dl4vc_amd.dataset import assemble_batch, select_rows
from dl4vc_amd.hdf5_schema import record_dtype


@pytest.fixture(scope="module")
def lib():
    if not loader.available():
        import __graft_entry__ as g
        g.build()
    return loader.load_library()


def test_symbols(lib):
    for s in loader.SYMBOLS:
        assert hasattr(lib, s)


def test_row_subset_matches_numpy_randomstate(lib):
    """The C++ MT19937 / random_sample / permutation reproduce numpy's legacy RandomState draw for draw."""
    out = (C.c_int32 * 200)()
    for seed, n, stored, mx in ((0, 150, 200, 100), (1, 101, 200, 100), (12345, 200, 200, 64), (7, 64, 200, 64),
                                (2 ** 31 + 5, 199, 200, 100), (99, 500, 200, 100)):
        k = lib.dl_select_rows(seed, n, stored, mx, out)
        rng = np.random.RandomState(seed)
        rng.random_sample()
        want = select_rows(n, stored, mx, rng)
        assert k == len(want) and list(out[:k]) == want.tolist(), (seed, n)


def test_allele_masks_match_reference_fixtures(lib):
    cases = [c for c in json.load(open(os.path.join(GOLDEN, "alleles.json"))) if "window" in c]
    for c in cases:
        win = np.array(c["window"], np.uint8)
        rm, vm = np.empty(201, np.uint8), np.empty(201, np.uint8)
        st = lib.dl_allele_masks(c["vcfrec"].encode(), win.ctypes.data, rm.ctypes.data, vm.ctypes.data)
        if c["error"] is None:
            assert st == 0 and rm.tolist() == c["ref_mask"] and vm.tolist() == c["var_mask"], c["vcfrec"]
        elif c["error"] == "AssertionError":
            assert st == 1 and rm.max() == 0 and vm.max() == 0          # blacklisted: zero masks
        else:
            assert st == 2                                             # the reference dies with a non-assert error


def test_batches_equal_python_assembly(lib, tmp_path):
    batch = synth.make_sites(70, reads=100, seed=13)
    recs = hdf5io.records_from_sites(batch)
    rng = np.random.default_rng(1)
    for i in (4, 33, 69):                                   # deep pileups -> seeded subsets
        recs[i]["num_reads"] = 100 + 10 * (i % 7) + 1
        recs[i]["single_reads"][100:] = rng.integers(0, 10, (100, 201))
        recs[i]["q-scores"][100:] = rng.integers(0, 42, (100, 201))
        recs[i]["strand"][100:] = rng.integers(0, 3, (100, 201))
    path = str(tmp_path / "c.hdf")
    hdf5io.write_candidates(path, recs, chunk=3)
    for R in (100, 64):
        got = []
        with loader.NativeLoader(path, reads=R, batch_sites=16, lo=5, hi=68, seed=1000, threads=4) as nl:
            assert len(nl) == 63 and nl.num_records == 70
            for b in nl:
                got.append(b)
        assert [len(b) for b in got] == [16, 16, 16, 15]
        want = assemble_batch(recs[5:68], R, seed=1000 + 5)
        for k in ("reads", "qual", "strand", "ref", "ref_mask", "var_mask", "num_reads"):
            np.testing.assert_array_equal(np.concatenate([getattr(b, k) for b in got]), getattr(want, k), err_msg=k)
        assert sum((b.vcfrec for b in got), []) == want.vcfrec
    # without a seed a deep pileup is refused, as in the Python path
    with loader.NativeLoader(path, reads=100, batch_sites=16, threads=2) as nl:
        with pytest.raises(ValueError, match="seed"):
            list(nl)


def test_very_deep_pileup_is_refused_by_both_paths(lib, tmp_path):
    """num_reads > 2 x stored rows: the sampling window [start, start + 200) runs past the 200 stored rows (450 reads ->
    start 125 -> 75 rows).  The reference yields a (201, 75) item its DataLoader cannot collate (dataset.py:517-521,
    :270-281); here both loaders refuse and name the record instead of zero-padding silently."""
    batch = synth.make_sites(5, reads=100, seed=3)
    recs = hdf5io.records_from_sites(batch)
    recs[3]["num_reads"] = 450
    path = str(tmp_path / "deep.hdf")
    hdf5io.write_candidates(path, recs)
    with pytest.raises(ValueError, match="75 stored rows"):
        assemble_batch(recs, 100, seed=5)
    with loader.NativeLoader(path, reads=100, batch_sites=4, seed=5, threads=2) as nl:
        with pytest.raises(ValueError, match="record 3: num_reads 450 leaves 75 stored rows"):
            list(nl)
    # R <= the rows that remain: both paths take the first... no: 450 > 64 draws a seeded subset of the 75 rows, identically
    with loader.NativeLoader(path, reads=64, batch_sites=8, seed=5, threads=2) as nl:
        (b,) = list(nl)
    want = assemble_batch(recs, 64, seed=5)
    np.testing.assert_array_equal(b.reads, want.reads)


def test_reference_dataset_fixture_through_native_loader(lib, tmp_path):
    """tests/golden/dataset_a2.npz was produced by the reference's ContextDatasetFromNumpy (np.random.seed(1000+i))."""
    z = np.load(os.path.join(GOLDEN, "dataset_a2.npz"))
    raw = np.load(os.path.join(GOLDEN, "records_a2.npz"))["records"]
    recs = raw.reshape(-1).view(record_dtype())
    path = str(tmp_path / "a2.hdf")
    hdf5io.write_candidates(path, recs)
    with loader.NativeLoader(path, reads=100, batch_sites=6, seed=1000, threads=1) as nl:
        (b,) = list(nl)
    meta = json.loads(bytes(z["meta_json"]).decode())
    for i in range(6):
        for k in ("reads", "qual", "strand", "ref", "ref_mask", "var_mask"):
            np.testing.assert_array_equal(getattr(b, k)[i], z["%s%d" % (k, i)], err_msg="%s %d" % (k, i))
        assert b.vcfrec[i] == meta[i]["vcfrec"] and bool(b.blacklist[i]) == meta[i]["blacklist"]


def test_native_loader_is_faster_than_python(lib, tmp_path):
    batch = synth.tile_sites(synth.make_sites(50, reads=64, seed=2), 400)
    recs = hdf5io.records_from_sites(batch)
    path = str(tmp_path / "speed.hdf")
    hdf5io.write_candidates(path, recs, chunk=2)
    t0 = time.perf_counter()
    with hdf5io.CandidateFile(path) as f:
        py = assemble_batch(f.read(0, 400), 64)
    t_py = time.perf_counter() - t0
    t0 = time.perf_counter()
    with loader.NativeLoader(path, reads=64, batch_sites=100, threads=4) as nl:
        n = sum(len(b) for b in nl)
    t_nat = time.perf_counter() - t0
    assert n == 400 and len(py) == 400
    print("python %.0f sites/s, native %.0f sites/s" % (400 / t_py, 400 / t_nat))
    assert t_nat < t_py
