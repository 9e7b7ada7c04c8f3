"""Fuzzing against the LIVE reference (only where /root/reference exists, i.e. the build container; skipped
on the GPU box).  Complements the committed golden vectors: random structural configurations through the
oracle vs the reference model, random allele records through the mask builders, random scored VCFs through
format_vcf."""
import contextlib
import io
import os
import random
import sys
import types

import numpy as np
import pytest

from conftest import ROOT

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gen_golden as G
    m, d, u = G.import_reference()
    return G, m, d, u


def test_random_structural_configs_oracle_equals_reference(ref):
    from oracle.dan_oracle import OracleSpec, random_state_dict, dan_forward_oracle
    from dl4vc_amd import synth
    G, m, d, u = ref
    rng = random.Random(5)
    for trial in range(9):
        layers = rng.choice([3, 4, 5, 7])
        pools = tuple(sorted(rng.sample(range(1, layers), rng.choice([0, 1, 2]) if layers > 2 else 0)))
        # a third of the trials above the reference's MAX_READS = 100 (gen_golden.build_reference_model raises the constant in the
        # imported module for those: dl4vc/model.py:12,194,303-304)
        spec = OracleSpec(reads=(rng.choice([101, 117, 128]) if trial % 3 == 1 else rng.choice([3, 5, 8])), c_init=rng.choice([4, 8, 12]), c_final=rng.choice([4, 8, 12]),
                          layers=layers, pool_layers=pools, residual_start=rng.choice([0, 2, 3]),
                          dil_mid=rng.choice([1, 2, 3]), dil_final=rng.choice([1, 2]), use_bn=rng.random() < 0.7,
                          use_q=rng.random() < 0.7, use_strand=rng.random() < 0.7, use_mask=rng.random() < 0.8,
                          bottleneck=rng.choice([0, 2, 4]), fc_sizes=(rng.choice([8, 12]), rng.choice([4, 6])))
        if spec.residual_start > layers:
            continue
        sd = random_state_dict(spec, seed=300 + trial)
        batch = synth.make_sites(3, reads=spec.reads, seed=400 + trial)
        want = G.run_reference(m, spec, sd, batch, taps=False)
        got = dan_forward_oracle(sd, spec, *batch.arrays())
        for k, v in want.items():
            np.testing.assert_allclose(got[k], v, atol=2e-5 * max(1.0, float(np.abs(v).max())), err_msg="%s %s" % (spec, k))


def test_bf16_operand_mode_of_the_oracle_equals_the_reference_run_with_bf16_rounded_gemm_operands(ref):
    """BASELINE config 5 computes in bf16.  The reference has no such mode; what it means is pinned here: the reference's OWN
    forward with the weights of every conv / residual 1x1 / bottleneck module rounded to bf16 and the inputs of those modules
    rounded to bf16 by forward-pre-hooks (fp32 sums, everything else untouched) equals the oracle's bf16 = "operands" mode --
    on random structures and on the production structure at 301 columns.  The "storage" mode (the HIP kernel's bf16 activation
    storage on top) stays within the bf16 bound of it."""
    from oracle.dan_oracle import OracleSpec, random_state_dict, dan_forward_oracle
    from dl4vc_amd import synth
    G, m, d, u = ref
    rng = random.Random(11)
    specs = [OracleSpec(reads=6, length=301, c_init=16, c_final=16, bottleneck=4, fc_sizes=(16, 8)),
             OracleSpec(reads=109, length=240, c_init=8, c_final=8, bottleneck=4, fc_sizes=(8, 4), pool_layers=(2, 4))]   # > MAX_READS
    for trial in range(4):
        layers = rng.choice([3, 5, 7])
        pools = tuple(sorted(rng.sample(range(1, layers), rng.choice([0, 1]))))
        specs.append(OracleSpec(reads=rng.choice([3, 5]), c_init=rng.choice([8, 12]), c_final=rng.choice([8, 12]), layers=layers,
                                pool_layers=pools, residual_start=rng.choice([0, 2, 3]), dil_mid=rng.choice([1, 2]),
                                dil_final=2, use_bn=rng.random() < 0.7, bottleneck=rng.choice([0, 4]), fc_sizes=(8, 4)))
    for i, spec in enumerate(specs):
        if spec.residual_start > spec.layers:
            continue
        sd = random_state_dict(spec, seed=700 + i)
        batch = synth.make_sites(3, reads=spec.reads, length=spec.length, seed=800 + i)
        want = G.run_reference(m, spec, sd, batch, taps=True, bf16_operands=True)
        got = dan_forward_oracle(sd, spec, *batch.arrays(), taps=True, bf16="operands")
        plain = dan_forward_oracle(sd, spec, *batch.arrays(), taps=True)
        stor = dan_forward_oracle(sd, spec, *batch.arrays(), taps=True, bf16="storage")
        moved = 0.0
        for k, v in want.items():
            sc = max(1.0, float(np.abs(v).max()))
            # the same arithmetic on the same torch kernels: the fp32-vs-fp32 bar (observed: conv taps bit-identical)
            np.testing.assert_allclose(got[k], v, atol=2e-5 * sc, err_msg="%s %s" % (spec, k))
            moved = max(moved, float(np.abs(plain[k] - v).max()) / sc)
            assert float(np.abs(stor[k] - v).max()) <= 4e-2 * sc, (spec, k)
        assert moved > 1e-4, "the hooks rounded nothing: %g" % moved      # the mode IS different from fp32


def test_random_alleles_equal_reference(ref):
    from dl4vc_amd import alleles
    G, m, d, u = ref
    rng = np.random.default_rng(9)
    bases = "ACGT"
    n_ok = n_err = 0
    for trial in range(300):
        window = rng.integers(1, 5, 201).astype(np.uint8)
        for g in rng.integers(95, 115, rng.integers(0, 4)):
            window[g] = 5                                           # gap columns around the centre
        kind = rng.integers(0, 4)
        c = "ATGC"[window[100] - 1] if window[100] in (1, 2, 3, 4) else "A"
        tok2chr = {1: "A", 2: "T", 3: "G", 4: "C", 5: "N"}
        if kind == 0:
            ref_s, alt_s = c, bases[rng.integers(0, 4)]
        elif kind == 1:
            k = int(rng.integers(1, 6))
            ref_s, alt_s = c, c + "".join(bases[i] for i in rng.integers(0, 4, k))
        elif kind == 2:
            k = int(rng.integers(1, 6))
            tail = [tok2chr[int(t)] for t in window[101:140] if t != 5][:k]      # the deleted reference bases
            ref_s, alt_s = c + "".join(tail), c
        else:
            ref_s = "".join(bases[i] for i in rng.integers(0, 4, rng.integers(1, 4)))
            alt_s = "".join(bases[i] for i in rng.integers(0, 4, rng.integers(1, 4)))
        rec = "\t".join(("chr1", "77", ".", ref_s, alt_s, "50", ".", "DP=10;AF=0.5", "GT:GQ", "1:50"))
        with contextlib.redirect_stdout(io.StringIO()):
            try:
                want = d.get_read_mask_vectors(rec, reference=window.copy())
                err = None
            except Exception as e:          # noqa: BLE001
                want, err = None, e
        if err is None:
            got = alleles.allele_mask_vectors(rec, window)
            assert got[0].tolist() == want[0].tolist() and got[1].tolist() == want[1].tolist(), rec
            n_ok += 1
        else:
            with pytest.raises(Exception) as ex:
                alleles.allele_mask_vectors(rec, window)
            assert isinstance(ex.value, AssertionError) == isinstance(err, AssertionError), (rec, err, ex.value)
            n_err += 1
    assert n_ok > 100 and n_err > 10


def test_random_scored_vcfs_through_format_vcf_equal_reference(ref, tmp_path):
    import importlib
    from dl4vc_amd import vcf
    with contextlib.redirect_stdout(io.StringIO()):
        fv = importlib.import_module("format_vcf")
    assert fv.__file__.startswith(REF)
    rng = np.random.default_rng(3)
    header = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tCALLED\n"
    alleles_pool = [("A", "G"), ("C", "T"), ("AT", "A"), ("A", "AT"), ("ATG", "A"), ("A", "ATGC"), ("G", "C"), ("T", "A")]
    for trial in range(25):
        lines, pos, used = [], 100, set()
        for _ in range(int(rng.integers(1, 40))):
            if rng.random() < 0.6:
                pos += int(rng.integers(1, 50))
                used = set()
            cands = [a for a in alleles_pool if a not in used]      # identical lines trip an assert in the reference
            if not cands:
                continue
            ref_s, alt_s = cands[int(rng.integers(0, len(cands)))]
            used.add((ref_s, alt_s))
            p = rng.dirichlet((0.6, 0.6, 0.6)) if rng.random() < 0.7 else np.array([rng.random() * 0.2, 0.05, 0.0])
            nv, hv, ov = (float(x) for x in p / max(p.sum(), 1e-9)) if p.sum() > 0 else (1.0, 0.0, 0.0)
            lines.append("\t".join(("chr2", str(pos), "BP=%.8f;NV=%.8f;HV=%.8f;OV=%.8f" % (1 - nv, nv, hv, ov), ref_s, alt_s,
                                    "50", ".", "DP=30;AF=0.5", "GT:GQ", "1:50")))
        text = header + "\n".join(lines) + "\n"
        pin, pout = str(tmp_path / ("in%d.vcf" % trial)), str(tmp_path / ("out%d.vcf" % trial))
        open(pin, "w").write(text)
        a = types.SimpleNamespace(input_file=pin, output_file=pout, snp_threshold=0.1, indel_threshold=0.2,
                                  long_indel_threshold=0.0, delete_threshold=0.0, snp_zygo_threshold=0.75,
                                  indel_zygo_threshold=0.8, long_indel_zygo_threshold=0.5, delete_zygo_threshold=0.5,
                                  multiallele_second_threshold=0.7, multiallele_homozygous_second_threshold=0.9, debug=False)
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            fv.filter_format_vcf(a)
        got = "".join(vcf.format_vcf_lines(text.splitlines(keepends=True), vcf.FormatOptions(**vcf.PIPELINE_OPTIONS)))
        assert got == open(pout).read(), "trial %d" % trial


def test_acceptance_tool_derives_the_genotype_lines_the_reference_program_writes(ref, tmp_path):
    """tools/compare_calls.py judges the real-data run.  Its genotype lines must be what the reference's
    OWN tools/format_vcf.py writes from the same scored file: random pairs of scored VCFs (the second = the first with score noise of
    a few 1e-5, some sites pushed across a threshold) go through the reference program, the two outputs are diffed record by record,
    and dl4vc_amd.compare must report exactly that set of differing records -- split into knife-edge and elsewhere by ITS attribution,
    which is then checked against the scores."""
    import importlib
    from dl4vc_amd import vcf
    from dl4vc_amd.compare import compare_scored_vcfs
    with contextlib.redirect_stdout(io.StringIO()):
        fv = importlib.import_module("format_vcf")
    rng = np.random.default_rng(17)
    header = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tCALLED\n"
    alleles_pool = [("A", "G"), ("C", "T"), ("AT", "A"), ("A", "AT"), ("ATG", "A"), ("A", "ATGC"), ("G", "C"), ("T", "A")]

    def run_reference(text, tag):
        pin, pout = str(tmp_path / (tag + ".in.vcf")), str(tmp_path / (tag + ".out.vcf"))
        open(pin, "w").write(text)
        a = types.SimpleNamespace(input_file=pin, output_file=pout, snp_threshold=0.1, indel_threshold=0.2,
                                  long_indel_threshold=0.0, delete_threshold=0.0, snp_zygo_threshold=0.75,
                                  indel_zygo_threshold=0.8, long_indel_zygo_threshold=0.5, delete_zygo_threshold=0.5,
                                  multiallele_second_threshold=0.7, multiallele_homozygous_second_threshold=0.9, debug=False)
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            fv.filter_format_vcf(a)
        out = {}
        for line in open(pout):
            if not line.startswith("#"):
                c = line.rstrip("\n").split("\t")
                out[(c[0], c[1], c[3], c[4])] = c[9].split(":")[0]
        return out

    total_diff = 0
    for trial in range(12):
        recs, pos, used = [], 100, set()
        for _ in range(int(rng.integers(20, 60))):
            if rng.random() < 0.7:
                pos += int(rng.integers(1, 50))
                used = set()
            cands = [a for a in alleles_pool if a not in used]
            if not cands:
                continue
            ref_s, alt_s = cands[int(rng.integers(0, len(cands)))]
            used.add((ref_s, alt_s))
            p = rng.dirichlet((0.6, 0.6, 0.6))
            if rng.random() < 0.25:                                  # plant the call score next to the SNP / indel threshold
                thr = 0.1 if (len(ref_s) == 1 and len(alt_s) == 1) else 0.2
                nv = 1.0 - thr + rng.uniform(-3e-5, 3e-5)
                p = np.array([nv, (1 - nv) * 0.7, (1 - nv) * 0.3])
            recs.append((pos, ref_s, alt_s, p))

        def text_of(scores):
            return header + "".join("\t".join(("chr2", str(pos), "BP=%.8f;NV=%.8f;HV=%.8f;OV=%.8f" % (1 - s[0], s[0], s[1], s[2]), r, al,
                                               "50", ".", "DP=30;AF=0.5", "GT:GQ", "1:50")) + "\n" for (pos, r, al, _), s in zip(recs, scores))
        sa = [p for _, _, _, p in recs]
        sb = []
        for p in sa:
            d = rng.normal(0, 2e-5, 3)
            q = np.clip(p + d - d.mean(), 0, 1)
            sb.append(q)
        ta, tb = text_of(sa), text_of(sb)
        # (the pipeline sorts the scored file before format_vcf -- call_variants.sh:151, `sort -k1,1 -k2,2n`, ties by the whole line --
        # and so does the tool; the reference program gets the sorted text, as it does in the pipeline)
        srt = lambda t: "".join(vcf.sort_scored_vcf_lines(t.splitlines(keepends=True)))      # noqa: E731
        ga, gb = run_reference(srt(ta), "a%d" % trial), run_reference(srt(tb), "b%d" % trial)
        want = {k for k in set(ga) | set(gb) if ga.get(k) != gb.get(k)}
        rep = compare_scored_vcfs(ta.splitlines(keepends=True), tb.splitlines(keepends=True), vcf.FormatOptions(**vcf.PIPELINE_OPTIONS), max_listed=1000)
        got = set()
        for kind in ("knife_edge", "elsewhere"):
            for e in rep["genotype_differences"][kind]["first"]:
                m = e["site"]
                chrom_pos, alle = m.split(" ")
                chrom, pos = chrom_pos.split(":")
                r, al = alle.split(">")
                got.add((chrom, pos, r, al))
        assert got == want, (trial, got ^ want)
        # every difference between two files 2e-5 apart is a knife edge by the tool's own attribution
        assert rep["genotype_differences"]["elsewhere"]["count"] == 0 and rep["ok"], rep["genotype_differences"]["elsewhere"]
        total_diff += len(want)
    assert total_diff >= 5, "the trials produced too few threshold crossings to test anything: %d" % total_diff
