"""Scored-VCF writer and genotype post-processing (rows A16/A17) against the reference's outputs."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from dl4vc_amd import vcf

FX = json.load(open(os.path.join(GOLDEN, "vcf.json")))


def test_append_scored_records_formatting(tmp_path):
    p = tmp_path / "epoch1_x.vcf"
    p.write_text("##fileformat=VCFv4.2\n")
    bp = np.array(FX["bp"], np.float32)
    vt = np.array(FX["vt"], np.float32)
    vcf.append_scored_records(str(p), bp, vt, FX["records"])
    assert p.read_text() == FX["file"]


def test_refuses_to_overwrite_id_column():
    with pytest.raises(AssertionError):
        vcf.scored_record("c\t1\trs1\tA\tG\t50\t.\tDP=1;AF=1\tGT\t1", 0.5, (0.1, 0.2, 0.7))


def test_output_naming(tmp_path):
    assert vcf.scored_vcf_path("/x/y/model_test.vcf") == "/x/y/epoch1_model_test.vcf"
    src = tmp_path / "cand.vcf"
    src.write_text("##a\n#CHROM\tPOS\nchr1\t1\t.\n")
    out = vcf.start_scored_vcf(str(src), str(tmp_path / "model_test.vcf"))
    assert os.path.basename(out) == "epoch1_model_test.vcf"
    assert open(out).read() == "##a\n#CHROM\tPOS\n"


@pytest.mark.parametrize("name", sorted(FX["format_vcf"]))
def test_format_vcf_scenarios(name):
    sc = FX["format_vcf"][name]
    opts = vcf.FormatOptions(**vcf.PIPELINE_OPTIONS)
    got = "".join(vcf.format_vcf_lines(sc["input"].splitlines(keepends=True), opts))
    assert got == sc["output"]


def test_two_base_delete_without_indel_threshold_is_a_nameerror():
    line = "c\t1\tBP=0.9;NV=0.1;HV=0.1;OV=0.8\tAT\tA\t50\t.\tDP=1;AF=1\tGT\t1\n"
    with pytest.raises(NameError):
        vcf.format_vcf_lines([line], vcf.FormatOptions())


def test_sort_stage_matches_gnu_sort(tmp_path):
    """call_variants.sh:151 pipes records through `sort -k1,1 -k2,2n`; ours must order identically (C locale)."""
    import random
    import shutil
    import subprocess
    if shutil.which("sort") is None or shutil.which("awk") is None:
        pytest.skip("coreutils not available")
    rng = random.Random(4)
    chroms = ["chr1", "chr10", "chr2", "chrX", "chr20", "1", "chr1_random"]
    lines = ["##fileformat=VCFv4.2\n", "#CHROM\tPOS\tID\n"]
    for _ in range(400):
        lines.append("\t".join((rng.choice(chroms), str(rng.choice([5, 50, 500, 1234567, 7])), "BP=%.8f" % rng.random(),
                                rng.choice("ACGT"), rng.choice("ACGT"))) + "\n")
    src = tmp_path / "in.vcf"
    src.write_text("".join(lines))
    cmd = "awk '$1 ~ /^#/ {print $0;next} {print $0 | \"sort -k1,1 -k2,2n\"}' %s" % src
    want = subprocess.run(["bash", "-c", cmd], capture_output=True, text=True, env={"LC_ALL": "C", "PATH": os.environ["PATH"]}).stdout
    assert "".join(vcf.sort_scored_vcf_lines(lines)) == want
