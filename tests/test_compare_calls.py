"""tools/compare_calls.py / dl4vc_amd/compare.py: the acceptance comparison of two scored VCFs (VERDICT r5 "next" 6; reference:
tools/format_vcf.py:92-221, dl4vc/utils.py:162-178, dl4vc/dataset.py:271-281).  CPU only: two synthetic files with a knife-edge site, a
real difference, a > 100-read site and a multi-allele position."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT
from dl4vc_amd.compare import compare_scored_vcfs
from dl4vc_amd.vcf import FormatOptions, PIPELINE_OPTIONS, scored_record

HEADER = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tCALLED\n"


def rec(chrom, pos, ref, alt):
    return "\t".join((chrom, str(pos), ".", ref, alt, "50", ".", "DP=30;AF=0.5", "GT:GQ", "1:50"))


def scored(sites, scores):
    """scores: (NV, HV, OV) per site; BP = 1 - NV (trainer.py:620-623 with the binary head tied for the test)"""
    return [HEADER] + [scored_record(r, 1.0 - s[0], s) + "\n" for r, s in zip(sites, scores)]


SITES = [rec("chr20", 100, "A", "G"), rec("chr20", 200, "A", "G"), rec("chr20", 300, "AT", "A"), rec("chr20", 400, "C", "T"),
         rec("chr20", 500, "A", "G"), rec("chr20", 500, "A", "C"), rec("chr20", 500, "A", "T"), rec("chr20", 600, "G", "GA")]
# SNP call threshold 0.1 on 1 - NV, homozygous 0.75 on OV; indel 0.2 / 0.8 (call_variants.sh:154-160)
BASE = [(0.05, 0.85, 0.10),          # called, 0/1
        (0.90002, 0.05, 0.04998),    # 1 - NV = 0.09998: NOT called, 2e-5 from the call threshold (knife edge)
        (0.35, 0.05, 0.60),          # deletion: called 0/1
        (0.02, 0.08, 0.90),          # called 1/1
        (0.20, 0.70, 0.10), (0.25, 0.65, 0.10), (0.60, 0.30, 0.10),     # three alleles at one position
        (0.50, 0.40, 0.10)]          # insertion called 0/1


def test_identical_files_agree():
    rep = compare_scored_vcfs(scored(SITES, BASE), scored(SITES, BASE))
    assert rep["ok"] and rep["calls_identical"] and rep["records_a"] == 8 and rep["sites_within_tolerance_of_a_threshold"] == 1
    assert all(v["max_abs_diff"] == 0.0 for v in rep["scores"].values())


def test_a_knife_edge_flip_is_attributed_and_tolerated_a_real_difference_is_not():
    other = [tuple(x) for x in BASE]
    other[1] = (0.89997, 0.05, 0.05003)                       # 1 - NV = 0.10003: called in B, 5e-5 away in scores
    rep = compare_scored_vcfs(scored(SITES, BASE), scored(SITES, other))
    g = rep["genotype_differences"]
    assert g["knife_edge"]["count"] == 1 and g["elsewhere"]["count"] == 0 and rep["ok"] and not rep["calls_identical"]
    assert g["knife_edge"]["first"][0]["a"] is None and g["knife_edge"]["first"][0]["b"].startswith("0/1")
    assert rep["scores"]["NV"]["max_abs_diff"] <= 1e-4
    # a genotype that differs far from any threshold (and a score 0.3 apart): DIFFERENT
    other[3] = (0.02, 0.38, 0.60)
    rep = compare_scored_vcfs(scored(SITES, BASE), scored(SITES, other))
    assert not rep["ok"] and rep["genotype_differences"]["elsewhere"]["count"] == 1
    assert rep["genotype_differences"]["elsewhere"]["first"][0]["site"] == "chr20:400 C>T"
    assert rep["scores"]["OV"]["beyond_tolerance"] == 1


def test_scores_beyond_the_tolerance_fail_even_when_every_call_agrees():
    other = [tuple(x) for x in BASE]
    other[0] = (0.0503, 0.8497, 0.10)                          # 3e-4 off, same call
    rep = compare_scored_vcfs(scored(SITES, BASE), scored(SITES, other))
    assert rep["calls_identical"] and not rep["ok"] and rep["scores"]["NV"]["beyond_tolerance"] == 1


def test_sites_with_more_reads_than_the_reference_keeps_are_set_apart():
    other = [tuple(x) for x in BASE]
    other[3] = (0.02, 0.38, 0.60)                              # the real difference of above ...
    nr = [40, 60, 80, 150, 30, 30, 30, 99]                     # ... on a 150-read site: the reference itself is random there
    rep = compare_scored_vcfs(scored(SITES, BASE), scored(SITES, other), num_reads=nr)
    assert rep["ok"] and rep["sites_with_more_reads_than_the_reference_keeps"] == 1 and rep["sites_deterministic"] == 7
    assert rep["genotype_differences"]["random_subset_sites"]["count"] == 1 and rep["scores"]["OV"]["beyond_tolerance"] == 0
    assert rep["scores"]["OV"]["max_abs_diff_random_subset_sites"] > 0.2
    nr[3] = 100                                                # exactly MAX_READS: deterministic (dataset.py:271: "> max_reads")
    assert not compare_scored_vcfs(scored(SITES, BASE), scored(SITES, other), num_reads=nr)["ok"]


def test_multi_allele_position_is_judged_as_a_group():
    """The pruning at a position with several called alleles ranks the group (format_vcf.py:158-196): a 6e-5 change of ONE allele's
    score across the second-allele threshold (0.7 on the call score) changes which OTHER alleles survive -- attributed to the knife
    edge through the group, not reported as a real difference of the allele that vanished."""
    a = [tuple(x) for x in BASE]
    b = [tuple(x) for x in BASE]
    a[4], a[5], a[6] = (0.10, 0.80, 0.10), (0.29997, 0.60, 0.10003), (0.60, 0.30, 0.10)    # second-best call score 0.70003 (> 0.7: two kept)
    b[4], b[5], b[6] = (0.10, 0.80, 0.10), (0.30003, 0.60, 0.09997), (0.60, 0.30, 0.10)    # 0.69997 (<= 0.7: only the best kept)
    rep = compare_scored_vcfs(scored(SITES, a), scored(SITES, b))
    g = rep["genotype_differences"]
    assert g["knife_edge"]["count"] >= 1 and g["elsewhere"]["count"] == 0 and rep["ok"], g


def test_missing_record_fails_and_cli_exit_codes(tmp_path):
    pa, pb, pc = tmp_path / "a.vcf", tmp_path / "b.vcf", tmp_path / "c.vcf"
    pa.write_text("".join(scored(SITES, BASE)))
    pb.write_text("".join(scored(SITES, BASE)))
    pc.write_text("".join(scored(SITES[:-1], BASE[:-1])))
    tool = os.path.join(ROOT, "tools", "compare_calls.py")
    r = subprocess.run([sys.executable, tool, str(pa), str(pb), "--json", str(tmp_path / "r.json")], capture_output=True, text=True)
    assert r.returncode == 0 and "RESULT: identical within the bars" in r.stdout, r.stdout + r.stderr
    assert os.path.isfile(tmp_path / "r.json")
    r = subprocess.run([sys.executable, tool, str(pa), str(pc)], capture_output=True, text=True)
    assert r.returncode == 1 and "only in A: 1" in r.stdout


def test_with_a_candidates_file(tmp_path):
    """--candidates: num_reads read from the HDF5 schema the scoring runs read (dl4vc/dataset.py)."""
    from dl4vc_amd import synth, hdf5io
    from dl4vc_amd.compare import read_num_reads
    batch = synth.make_sites(5, reads=100, seed=3)
    recs = hdf5io.records_from_sites(batch)
    recs["num_reads"] = np.array([20, 101, 100, 180, 7], np.int32)
    path = str(tmp_path / "candidates.hdf")
    hdf5io.write_candidates(path, recs)
    assert read_num_reads(path).tolist() == [20, 101, 100, 180, 7]
