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


def test_threshold_distance_uses_the_records_allele_class():
    """dl4vc_amd.vcf.threshold_distance: SNP records are held against 0.1 / 0.75, short indels against 0.2 / 0.8 (call_variants.sh:154-160)."""
    import numpy as np
    from dl4vc_amd import vcf
    opts = vcf.FormatOptions(**vcf.PIPELINE_OPTIONS)
    snp = "chr1\t100\t.\tA\tC\t50\t.\tDP=8\tGT\t0/1"
    ins = "chr1\t200\t.\tA\tAT\t50\t.\tDP=8\tGT\t0/1"
    vt = np.array([[0.90005, 0.05, 0.04995],      # SNP: call score 0.09995 -> 5e-5 under the 0.1 threshold
                   [0.90005, 0.05, 0.04995],      # insert: 0.2 is 0.1 away
                   [0.10, 0.15, 0.75002],         # SNP: OV 2e-5 above 0.75; call score 0.9 = the homozygous multi-allele threshold
                   [0.05, 0.15, 0.80]])           # insert: OV on 0.8
    d = vcf.threshold_distance([snp, ins, snp, ins], vt, opts)
    assert abs(d[0] - 5e-5) < 1e-9 and abs(d[1] - 0.10005) < 1e-9 and d[2] < 1e-9 and d[3] < 1e-12
