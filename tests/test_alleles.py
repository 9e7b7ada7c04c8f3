"""Allele masks / VCF parsing (rows A2-A3) against outputs of the reference's own functions."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from dl4vc_amd import alleles
from dl4vc_amd.dataset import assemble_site
from dl4vc_amd.hdf5_schema import record_dtype

CASES = json.load(open(os.path.join(GOLDEN, "alleles.json")))
MASK_CASES = [c for c in CASES if "window" in c]


@pytest.mark.parametrize("i", range(len(MASK_CASES)))
def test_mask_vectors(i):
    c = MASK_CASES[i]
    window = np.array(c["window"], np.uint8)
    if c["error"] is None:
        rm, vm = alleles.allele_mask_vectors(c["vcfrec"], window)
        assert rm.tolist() == c["ref_mask"] and vm.tolist() == c["var_mask"]
        assert rm.dtype == np.uint8
    else:
        with pytest.raises(Exception) as e:
            alleles.allele_mask_vectors(c["vcfrec"], window)
        # same error CLASS as the reference: AssertionError is what the dataset layer blacklists
        assert (c["error"] == "AssertionError") == isinstance(e.value, AssertionError)
        assert type(e.value).__name__ == c["error"] or isinstance(e.value, AssertionError)


@pytest.mark.parametrize("i", range(len(MASK_CASES)))
def test_token_vectors_and_parse(i):
    c = MASK_CASES[i]
    r, v = alleles.allele_token_vectors(c["vcfrec"])
    assert r.tolist() == c["ref_vec51"] and v.tolist() == c["var_vec51"]
    if isinstance(c["parse"], dict):
        assert alleles.parse_candidate(c["vcfrec"]) == c["parse"]
    if "count" in c:
        reads = np.array(c["count_reads"], np.uint8)
        got = alleles.count_center_support(reads, np.array(c["window"], np.uint8), c["parse"]["var_mode"])
        assert list(got) == c["count"]


def test_known_answers_from_survey():
    # SURVEY.md section 8a row A3 probe known-answers
    win = np.full(201, 4, np.uint8)
    rec = lambda r, a: "\t".join(("c", "1", ".", r, a, "50", ".", "DP=1;AF=1", "GT", "1"))  # noqa: E731
    win[100:103] = (1, 2, 3)
    rm, vm = alleles.allele_mask_vectors(rec("A", "G"), win)
    assert np.flatnonzero(rm).tolist() == [100] and (rm[100], vm[100]) == (1, 3)
    rm, vm = alleles.allele_mask_vectors(rec("ATG", "A"), win)
    assert rm[100:103].tolist() == [1, 2, 3] and vm[100:103].tolist() == [1, 5, 5]
    win2 = win.copy(); win2[101:103] = 5
    rm, vm = alleles.allele_mask_vectors(rec("A", "ATT"), win2)
    assert rm[100:103].tolist() == [1, 8, 8] and vm[100:103].tolist() == [1, 2, 2]
    win3 = win.copy(); win3[100:104] = (1, 5, 2, 3)
    rm, vm = alleles.allele_mask_vectors(rec("ATG", "A"), win3)
    assert np.flatnonzero(rm).tolist() == [100, 102, 103]


def test_truth_column_parse():
    for c in CASES:
        if "parse_only" in c:
            assert alleles.parse_candidate(c["vcfrec"]) == c["parse_only"]


def test_site_assembly_matches_reference_dataset():
    z = np.load(os.path.join(GOLDEN, "dataset_a2.npz"))
    raw = np.load(os.path.join(GOLDEN, "records_a2.npz"))["records"]
    recs = raw.reshape(-1).view(record_dtype())
    meta = json.loads(bytes(z["meta_json"]).decode())
    assert len(recs) == len(meta) == 6
    for i, m in enumerate(meta):
        site = assemble_site(recs[i], max_reads=100, rng=np.random.RandomState(m["seed"]))
        for k in ("reads", "qual", "strand", "ref", "ref_mask", "var_mask"):
            np.testing.assert_array_equal(getattr(site, k), z["%s%d" % (k, i)], err_msg="%s %d" % (k, i))
        assert site.vcfrec == m["vcfrec"] and site.name == m["name"]
        assert site.num_reads == m["num_reads"] and site.blacklist == m["blacklist"]
    # a deep pileup without an explicit generator is refused rather than silently unpinned
    deep = [i for i, m in enumerate(meta) if m["num_reads"] > 100][0]
    with pytest.raises(ValueError):
        assemble_site(recs[deep], max_reads=100, rng=None)
