"""The in-process tail of the pipeline (dl4vc_amd/vcfpost.py): multi-allele join, genotype rewrites, BGZF and tabix index.
The external tools it stands in for are absent from the build image, so these are known-answer tests of the documented
rules plus format round trips -- parity with bcftools/htslib output itself is unpinned (see the module docstring)."""
import gzip
import os
import random

from dl4vc_amd import vcfpost
from dl4vc_amd.vcf import FormatOptions, PIPELINE_OPTIONS, format_vcf_lines, score_field

HEADER = ["##fileformat=VCFv4.2\n",
          '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Genotype Quality">\n',
          '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n',
          '##INFO=<ID=DP,Number=1,Type=Integer,Description="Total Depth">\n',
          '##INFO=<ID=AF,Number=A,Type=Float,Description="Allele Frequency">\n',
          "##contig=<ID=chr20,length=64444167>\n", "##contig=<ID=chr21,length=46709983>\n",
          "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tCALLED\n"]


def rec(chrom, pos, ref, alt, gt, q=37, vid=".", qual="50", dp=30, af="0.5"):
    return "%s\t%d\t%s\t%s\t%s\t%s\t.\tDP=%d;AF=%s\tGT:GQ\t%s:%d\n" % (chrom, pos, vid, ref, alt, qual, dp, af, gt, q)


def body(lines):
    return [l.rstrip("\n").split("\t") for l in lines if not l.startswith("#")]


def test_join_two_het_snps_gives_0_2_then_sed_makes_1_2():
    lines = HEADER + [rec("chr20", 100, "A", "G", "0/1", 37, "id1", af="0.5"), rec("chr20", 100, "A", "T", "0/1", 20, "id2", dp=31, af="0.3"),
                      rec("chr20", 105, "C", "T", "1/1")]
    joined = vcfpost.join_multiallelic_lines(lines)
    assert joined[:len(HEADER)] == HEADER
    rows = body(joined)
    assert len(rows) == 2
    assert rows[0][:5] == ["chr20", "100", "id1;id2", "A", "G,T"]
    assert rows[0][7] == "DP=30;AF=0.5,0.3"                     # Number=1: first record; Number=A: per ALT
    assert rows[0][8:] == ["GT:GQ", "0/2:37"]                    # the reference's sed lines exist because of exactly this
    assert rows[1][4] == "T" and rows[1][9].startswith("1/1")
    final = body(vcfpost.genotype_rewrites(joined))
    assert final[0][9] == "1/2:37" and final[1][9] == "1/1:37"


def test_join_genotype_table():
    def gt_of(a, b):
        out = vcfpost.join_multiallelic_lines(HEADER + [rec("chr20", 7, "A", "G", a), rec("chr20", 7, "A", "T", b)])
        return body(vcfpost.genotype_rewrites(out))[0][9].split(":")[0], body(out)[0][9].split(":")[0]
    assert gt_of("0/1", "0/1") == ("1/2", "0/2")
    assert gt_of("1/1", "0/1") == ("1/2", "1/2")
    assert gt_of("0/1", "1/1") == ("1/2", "2/2")
    assert gt_of("1/1", "1/1") == ("1/2", "2/2")


def test_join_extends_alleles_to_the_longest_ref_and_merges_duplicates():
    lines = HEADER + [rec("chr20", 50, "A", "T", "0/1"), rec("chr20", 50, "ATG", "A", "0/1"), rec("chr20", 50, "AT", "TT", "0/1")]
    row = body(vcfpost.join_multiallelic_lines(lines))[0]
    assert row[3] == "ATG"
    assert row[4] == "TTG,A"                                     # SNP A>T == AT>TT once both are written on REF=ATG
    assert row[9].split(":")[0] == "0/1"                         # third record re-uses allele 1; second wrote 2 before it
    assert row[7] == "DP=30;AF=0.5,0.5"


def test_join_keeps_order_and_does_not_cross_chromosomes():
    lines = HEADER + [rec("chr20", 9, "A", "G", "0/1"), rec("chr21", 9, "A", "T", "0/1"), rec("chr21", 9, "A", "C", "1/1")]
    rows = body(vcfpost.join_multiallelic_lines(lines))
    assert [(r[0], r[1], r[4]) for r in rows] == [("chr20", "9", "G"), ("chr21", "9", "T,C")]


def test_sed_rewrites_are_first_occurrence_per_line():
    assert vcfpost.genotype_rewrites(["x\t0/2\t0/2\n", "2/2 2/2\n", "##note 0/2\n"]) == ["x\t1/2\t0/2\n", "1/2 2/2\n", "##note 1/2\n"]


def test_bgzf_is_valid_gzip_with_eof_block_and_small_blocks():
    random.seed(3)
    data = bytes(random.getrandbits(8) for _ in range(200000)) + b"A" * 100000
    blob = vcfpost.bgzf_compress(data)
    assert gzip.decompress(blob) == data
    assert blob.endswith(vcfpost.BGZF_EOF)
    sizes = [len(d) for _, d in vcfpost.bgzf_blocks(blob)]
    assert max(sizes) <= 0xff00 and sizes[-1] == 0 and sum(sizes) == len(data)
    assert vcfpost.bgzf_compress(b"") == vcfpost.BGZF_EOF


def test_tabix_index_round_trip(tmp_path):
    random.seed(11)
    lines = list(HEADER)
    truth = []
    for chrom, n in (("chr20", 4000), ("chr21", 1500)):
        pos = 1
        for _ in range(n):
            pos += random.choice([1, 3, 40, 700, 20000])
            ref = random.choice(["A", "AT", "ACGTACGTAC"])
            lines.append(rec(chrom, pos, ref, "G", "0/1"))
            truth.append((chrom, pos, pos + len(ref) - 1, lines[-1].rstrip("\n")))
    gz = str(tmp_path / "calls.vcf.gz")
    vcfpost.write_vcf_gz_with_index(lines, gz)
    assert gzip.open(gz, "rt").read() == "".join(lines)
    idx = vcfpost.read_tbi(gz + ".tbi")
    assert idx["names"] == ["chr20", "chr21"] and idx["format"] == 2 and idx["cols"] == (1, 2, 0) and idx["meta"] == "#"
    for chrom, b, e in [("chr20", 1, 1000), ("chr20", 16000, 17000), ("chr20", 1000000, 1200000), ("chr21", 5, 300000),
                        ("chr21", 10 ** 8, 10 ** 8 + 5), ("chrX", 1, 100)] + \
                       [(random.choice(["chr20", "chr21"]), s, s + w) for s, w in
                        ((random.randrange(1, 9000000), random.choice([1, 50, 20000, 400000])) for _ in range(40))]:
        want = [t[3] for t in truth if t[0] == chrom and t[1] <= e and t[2] >= b]
        assert vcfpost.tabix_query(gz, chrom, b, e) == want, (chrom, b, e)


def test_whole_tail_on_scored_records(tmp_path):
    """scored records -> format_vcf -> join -> rewrites -> .vcf.gz + .tbi, all in process."""
    def scored(pos, ref, alt, nv, ov):
        return "chr20\t%d\t%s\t%s\t%s\t50\t.\tDP=30;AF=0.5\tGT:GQ\t1:50\n" % (pos, score_field(1 - nv, (nv, 1 - nv - ov, ov)), ref, alt)
    lines = HEADER + [scored(100, "A", "G", 0.05, 0.1), scored(100, "A", "T", 0.2, 0.05), scored(200, "C", "T", 0.95, 0.0),
                      scored(300, "G", "GA", 0.1, 0.85), scored(400, "T", "C", 0.5, 0.2)]
    thres = format_vcf_lines(lines, FormatOptions(**PIPELINE_OPTIONS))
    p = tmp_path / "thres.vcf"
    p.write_text("".join(thres))
    vcfpost.finish_calls(str(p), str(tmp_path / "join.vcf"), str(tmp_path / "called_variants.vcf.gz"))
    rows = body(open(tmp_path / "join.vcf").readlines())
    assert [(r[1], r[4], r[9].split(":")[0]) for r in rows][:1] == [("100", "G,T", "1/2")]
    assert os.path.getsize(tmp_path / "called_variants.vcf.gz.tbi") > 0
    assert [l.split("\t")[1] for l in vcfpost.tabix_query(str(tmp_path / "called_variants.vcf.gz"), "chr20", 250, 350)] == ["300"]
