"""Pileup encoder: BAM / BAI / FASTA readers, pileup columns, and the image logic against
golden vectors computed by the reference's own helper functions (tests/golden/pileup_encoder.json.gz, made by
oracle/gen_golden_pileup.py).  CPU only."""
import gzip
import hashlib
import json
import os
import struct

import numpy as np
import pytest

from dl4vc_amd import bamio
from dl4vc_amd.bamio import (BamFile, BamWriter, FastaFile, build_bai, CMATCH, CINS, CDEL, CREF_SKIP, CSOFT_CLIP, CHARD_CLIP, CPAD,
                             FREVERSE, FUNMAP, FDUP, FSECONDARY)
from dl4vc_amd.pileup import pileup_columns
from dl4vc_amd import pileup_encoder as PE

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "pileup_encoder.json.gz")


def golden():
    with gzip.open(GOLDEN, "rt") as f:
        return json.load(f)


def check(arr, want, what):
    arr = np.ascontiguousarray(arr, np.uint8)
    assert list(arr.shape) == want["shape"], (what, arr.shape, want["shape"])
    if "values" in want:
        assert np.array_equal(arr, np.array(want["values"], np.uint8).reshape(want["shape"])), what
    assert hashlib.sha256(arr.tobytes()).hexdigest() == want["sha256"], what


# ------------------------------------------------------------------------------------------------------
# image logic: pinned by the reference's helpers
# ------------------------------------------------------------------------------------------------------
def test_decode_base_matches_reference_decode_base_detail():
    for c in golden()["decode"]:
        if "raises" in c:
            with pytest.raises(Exception):
                PE.decode_base(c["s"])
        else:
            got = PE.decode_base(c["s"])
            assert [int(v) if not isinstance(v, list) else [int(x) for x in v] for v in got] == c["want"], c["s"]


def test_images_and_records_match_reference_helpers():
    """Every case: the three planes of ``process_columns`` (bases, qualities, strands), the centre column, the column map and
    the cropped / padded record of ``finish_record`` equal what the reference's add_bases_to_alignment_image /
    handle_ended_sequences / resize / centre / trim functions produce on the same pileup columns -- incl. inserts longer than
    the cap, a capped-at-zero run, duplicated read ids sharing a row, more rows than the initial 1200 and a location whose
    own position has no column."""
    cases = golden()["cases"]
    assert len(cases) >= 10
    seen_none = False
    for c in cases:
        opt = PE.EncoderOptions(window_size=c["window_size"], max_reads=c["max_reads"], max_insert_length=c["max_insert_length"],
                                max_insert_length_variant=c["max_insert_length_variant"])
        table = c["id_table"]
        cols = [PE.ColumnInput(k["reference_pos"], k["sequences"], k["qualities"], [table[i] for i in k["ids"]], k["ref_base"])
                for k in c["columns"]]
        res = PE.process_columns(cols, c["center_position"], opt)
        if c["want"] is None:
            assert res is None
            seen_none = True
            continue
        img, center, colmap, q, s = res
        w = c["want"]
        assert center == w["center"]
        check(img, w["image"], "image")
        check(q, w["quality"], "quality")
        check(s, w["strand"], "strand")
        assert [[k, v[0], v[1], v[2]] for k, v in colmap.items()] == w["colmap"]
        dtype = PE.record_dtype(opt.max_reads, 2 * opt.window_size + 1)
        loc = PE.Location("ref", c["center_position"], "ref:%d" % c["center_position"], 2, "ref\t%d\t.\tA\tC" % c["center_position"])
        rec = PE.finish_record(res, loc, opt, dtype)
        if w["record"] is None:
            assert rec is None
            continue
        for k in ("single_reads", "ref_bases", "q-scores", "strand"):
            check(rec[k], w["record"][k], k)
        assert int(rec["num_reads"]) == w["record"]["num_reads"] and int(rec["label"]) == 2
        assert bytes(rec["name"]).rstrip(b"\x00") == loc.name.encode()[:16]
        assert not rec["ref"].any() and not rec["reads"].any()
    assert seen_none


# ------------------------------------------------------------------------------------------------------
# BAM / BAI / FASTA readers (specification layout; unpinned: no htslib in the image)
# ------------------------------------------------------------------------------------------------------
SPEC_READS = [   # the SAM specification's example alignment (section 1.1), 0-based positions
    ("r001", 99, 6, 30, [(CMATCH, 8), (CINS, 2), (CMATCH, 4), (CDEL, 1), (CMATCH, 3)], "TTAGATAAAGGATACTG"),
    ("r002", 0, 8, 30, [(CSOFT_CLIP, 3), (CMATCH, 6), (CPAD, 1), (CINS, 1), (CMATCH, 4)], "AAAAGATAAGGATA"),
    ("r003", 0, 8, 30, [(CSOFT_CLIP, 5), (CMATCH, 6)], "GCCTAAGCTAA"),
    ("r004", 0, 15, 30, [(CMATCH, 6), (CREF_SKIP, 14), (CMATCH, 5)], "ATAGCTTCAGC"),
    ("r003", 2064, 28, 17, [(CHARD_CLIP, 6), (CMATCH, 5)], "TAGGC"),
    ("r001", 147, 36, 30, [(CMATCH, 9)], "CAGCGGCAT"),
]


def write_spec_bam(path, extra=()):
    with BamWriter(path, [("ref", 45), ("other", 1000)]) as w:
        for name, flag, pos, mapq, cigar, seq in list(SPEC_READS) + list(extra):
            w.write(0, pos, name, flag, mapq, cigar, seq, [min(40, 10 + i) for i in range(len(seq))])


def test_bam_roundtrip_header_records_and_hand_packed_bytes(tmp_path):
    p = str(tmp_path / "spec.bam")
    write_spec_bam(p)
    with BamFile(p) as bam:
        assert bam.references == ["ref", "other"] and bam.lengths == [45, 1000]
        assert "@SQ\tSN:ref\tLN:45" in bam.header_text and bam.get_tid("chrref") == 0 and bam.get_tid("nope") == -1
        recs = list(bam)
    assert [(r.name, r.flag, r.pos, r.mapq, r.seq, r.cigar_string()) for r in recs] == [
        (n, f, p_, q, s, "".join("%d%s" % (l, bamio.CIGAR_OPS[op]) for op, l in c)) for n, f, p_, q, c, s in SPEC_READS]
    assert recs[0].reference_end == 22 and recs[3].reference_end == 40 and recs[4].is_reverse and recs[5].is_reverse and not recs[3].is_reverse
    assert recs[0].qual.tolist()[:3] == [10, 11, 12]
    # a record packed by hand from the specification's field table: refID 0, pos 6, l_read_name 5, mapq 30, bin 4681,
    # n_cigar 1, flag 16, l_seq 3, next -1/-1, tlen 0, "r00x\0", 3M = 3<<4|0, seq ACG = 0x12 0x40, qual 1 2 3
    raw = struct.pack("<iiBBHHHiiii", 0, 6, 5, 30, 4681, 1, 16, 3, -1, -1, 0) + b"r00x\x00" + struct.pack("<I", 48) + bytes([0x12, 0x40]) + bytes([1, 2, 3])
    r = bamio.parse_record(raw)
    assert (r.name, r.pos, r.flag, r.seq, r.cigar, r.qual.tolist(), r.is_reverse) == ("r00x", 6, 16, "ACG", ((CMATCH, 3),), [1, 2, 3], True)
    # the writer's own packing of the same record is byte-identical to the hand-packed one
    assert bamio.pack_record(0, 6, "r00x", 16, 30, [(CMATCH, 3)], "ACG", [1, 2, 3])[4:] == raw


def test_bgzf_multi_block_and_indexed_fetch_equals_linear_scan(tmp_path):
    rng = np.random.default_rng(5)
    p = str(tmp_path / "big.bam")
    starts = np.sort(rng.integers(0, 200000, 4000))
    with BamWriter(p, [("chr1", 250000), ("chr2", 50000)]) as w:
        for i, s in enumerate(starts):
            n = int(rng.integers(30, 151))
            w.write(0, int(s), "q%d" % i, 0, 60, [(CMATCH, n)], "".join(rng.choice(list("ACGT"), n)), rng.integers(2, 41, n).tolist())
        for i in range(50):
            w.write(1, 100 * i, "z%d" % i, 0, 60, [(CMATCH, 50)], "A" * 50, [30] * 50)
    assert os.path.getsize(p) > 3 * 65536                                  # several BGZF blocks
    with BamFile(p) as bam:
        assert bam.index is None
        everything = list(bam)
        assert len(everything) == 4050 and [r.pos for r in everything[:4000]] == starts.tolist()
        want = {}
        for lo, hi in ((0, 500), (16384, 16400), (99990, 100200), (199000, 260000), (240000, 250000)):
            want[(lo, hi)] = [r.name for r in bam.fetch(0, lo, hi)]
            assert want[(lo, hi)] == [r.name for r in everything if r.tid == 0 and r.pos < hi and r.reference_end > lo]
        assert len(list(bam.fetch(1, 0, 10))) == 1 and list(bam.fetch(5, 0, 10)) == []
    idx = build_bai(p, p + ".bai")
    assert len(idx.linear[0]) == (int(starts.max()) + 150 >> 14) + 1 or len(idx.linear[0]) >= (int(starts.max()) >> 14) + 1
    with BamFile(p) as bam:
        assert bam.index is not None
        for (lo, hi), names in want.items():
            assert [r.name for r in bam.fetch(0, lo, hi)] == names, (lo, hi)
        assert [r.name for r in bam.fetch(1, 4900, 4950)] == ["z49"]
    again = bamio.BaiIndex.load(p + ".bai")
    assert again.linear == idx.linear and again.bins == idx.bins
    with pytest.raises(ValueError, match="not a BGZF block"):
        BamFile(p + ".bai")


def test_fasta_fetch_with_and_without_fai(tmp_path):
    p = str(tmp_path / "ref.fa")
    seq1 = "ACGTTGCAAC" * 13 + "GGG"
    seq2 = "ttagcatN" * 9
    with open(p, "w") as f:
        f.write(">chr1 first\n" + "\n".join(seq1[i:i + 60] for i in range(0, len(seq1), 60)) + "\n")
        f.write(">2\n" + "\n".join(seq2[i:i + 50] for i in range(0, len(seq2), 50)) + "\n")
    fa = FastaFile(p)
    assert fa.references == ["chr1", "2"] and fa.get_reference_length("1") == len(seq1)
    for lo, hi in ((0, 1), (58, 63), (59, 60), (60, 61), (119, 130), (0, len(seq1)), (125, 500)):
        assert fa.fetch("chr1", lo, hi) == seq1[lo:hi]
        assert fa.fetch("1", lo, hi) == seq1[lo:hi]
    assert fa.fetch("chr2", 45, 55) == seq2[45:55] and fa.fetch("2", 70, 90) == seq2[70:72]
    with pytest.raises(KeyError):
        fa.fetch("chr3", 0, 1)
    with open(p + ".fai", "w") as f:                                     # the same index, from a .fai file
        for name, (length, off, lb, lw) in fa.index.items():
            f.write("%s\t%d\t%d\t%d\t%d\n" % (name, length, off, lb, lw))
    fb = FastaFile(p)
    assert fb.index == fa.index and fb.fetch("chr1", 55, 125) == seq1[55:125]


# ------------------------------------------------------------------------------------------------------
# pileup columns: the SAM specification's example, columns derived by hand
# ------------------------------------------------------------------------------------------------------
def spec_columns(**kw):
    recs = [bamio.parse_record(bamio.pack_record(0, pos, name, flag, mapq, cigar, seq, [min(40, 10 + i) for i in range(len(seq))])[4:])
            for name, flag, pos, mapq, cigar, seq in SPEC_READS]
    return {c.reference_pos: c for c in pileup_columns(recs, kw.pop("start", 0), kw.pop("stop", 45), **kw)}


def test_pileup_columns_of_the_sam_specification_example():
    cols = spec_columns()
    assert sorted(cols) == list(range(6, 45))                             # positions 7..45 (1-based) are covered, 1..6 are not
    q = {p: c.query_sequences() for p, c in cols.items()}
    assert q[6] == ["^?T"]                                                 # r001 starts; mapq 30 -> '?'
    assert q[8] == ["A", "^?A", "^?A"]                                     # r002 / r003 start behind their soft clips
    assert cols[8].entries[1].qpos == 3 and cols[8].entries[2].qpos == 5
    assert q[13] == ["A+2AG", "A+1G", "A$"]                                # r001 8M|2I, r002 6M 1P 1I, r003 ends
    assert q[17] == ["A-1N", "A$", "A"]                                    # r001 before its deletion; r002 ends; r004
    assert q[18] == ["*", "G"] and q[21] == ["G$", ">"]                    # deletion; r001 ends, r004 inside its 14N skip
    assert q[14] == ["G", "G"] and q[15] == ["A", "A", "^?A"] and q[20] == ["T", "T"]
    assert cols[21].entries[1].is_refskip and cols[21].entries[1].is_del and not cols[18].entries[0].is_refskip
    assert q[28] == [">", "^2t"] and q[32] == [">", "c$"]                  # supplementary r003 (flag 2064: reverse; mapq 17 -> '2')
    assert q[35] == ["T"] and q[36] == ["C", "^?c"] and q[39] == ["C$", "c"] and q[44] == ["t$"]   # r001/2 is reverse: lower case
    assert cols[18].query_qualities() == [10 + 14, 10 + 3]                 # a deletion reports the NEXT base's quality (r001: base 14)
    assert cols[13].query_ids()[0] == "r001:TTAGATAAAGGATACTG"
    # truncation to [start, stop) and the flag mask
    cut = spec_columns(start=10, stop=20)
    assert sorted(cut) == list(range(10, 20)) and cut[10].query_sequences() == ["A", "A", "C"]   # TTAG[A]TAA, AG[A]TAA, AG[C]TAA
    recs = [bamio.parse_record(bamio.pack_record(0, 5, "u", fl, 9, [(CMATCH, 4)], "ACGT", [9] * 4)[4:]) for fl in (FUNMAP, FDUP, FSECONDARY, 512)]
    assert list(pileup_columns(recs, 0, 45)) == []
    # min_base_quality drops entries from strings, qualities and ids alike
    c13 = cols[13]
    assert c13.query_qualities() == [17, 18, 20]
    assert c13.query_sequences(min_base_quality=19) == ["A$"] and c13.query_qualities(19) == [20] and c13.query_ids(19) == ["r003:GCCTAAGCTAA"]


def test_deletion_runs_merge_and_max_depth():
    recs = [bamio.parse_record(bamio.pack_record(0, 10, "d", 0, 20, [(CMATCH, 3), (CDEL, 1), (CDEL, 2), (CMATCH, 2)], "ACGTT", [30] * 5)[4:])]
    cols = {c.reference_pos: c.query_sequences() for c in pileup_columns(recs, 0, 100)}
    assert cols[12] == ["G-3NNN"] and cols[13] == ["*"] and cols[14] == ["*"] and cols[15] == ["*"] and cols[16] == ["T"]
    many = [bamio.parse_record(bamio.pack_record(0, 50, "m%d" % i, 0, 20, [(CMATCH, 5)], "ACGTA", [30] * 5)[4:]) for i in range(30)]
    assert len(next(iter(pileup_columns(many, 0, 100, max_depth=8))).entries) == 9     # the iterator stops taking reads at its position


# ------------------------------------------------------------------------------------------------------
# end to end: BAM + FASTA + VCF -> records the dataset layer accepts
# ------------------------------------------------------------------------------------------------------
def test_encode_locations_end_to_end(tmp_path):
    rng = np.random.default_rng(12)
    ref = "".join(rng.choice(list("ACGT"), 3000))
    fa = str(tmp_path / "ref.fa")
    open(fa, "w").write(">chr20\n" + "\n".join(ref[i:i + 70] for i in range(0, 3000, 70)) + "\n")
    bam = str(tmp_path / "reads.bam")
    snp_pos, ins_pos = 1500, 1900                                          # VCF POS (1-based)
    alt = "A" if ref[snp_pos - 1] != "A" else "C"
    reads = []
    for i in range(120):
        s = int(rng.integers(1200, 2000))
        n = 150
        seq = list(ref[s:s + n])
        cigar = [(CMATCH, n)]
        if s <= snp_pos - 1 < s + n and i % 2 == 0:
            seq[snp_pos - 1 - s] = alt
        if s < ins_pos - 1 < s + n - 1 and i % 3 == 0:
            k = ins_pos - s
            seq = seq[:k] + list("GGTT") + seq[k:]
            cigar = [(CMATCH, k), (CINS, 4), (CMATCH, n - k)]
        reads.append((s, "frag%d" % i, FREVERSE if i % 2 else 0, cigar, "".join(seq)))
    reads.sort()
    with BamWriter(bam, [("chr20", 3000)]) as w:
        for s, name, flag, cigar, seq in reads:
            w.write(0, s, name, flag, 60, cigar, seq, rng.integers(20, 41, len(seq)).tolist())
    build_bai(bam, bam + ".bai")
    vcf = str(tmp_path / "candidates.vcf")
    open(vcf, "w").write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
                         "chr20\t%d\t.\t%s\t%s\t50\t.\tDP=60;AF=0.5\n" % (snp_pos, ref[snp_pos - 1], alt) +
                         "chr20\t%d\t.\t%s\t%sGGTT\t50\t.\tDP=60;AF=0.33\n" % (ins_pos, ref[ins_pos - 1], ref[ins_pos - 1]) +
                         "chr20\t2990\t.\tA\tC\t50\t.\tDP=0;AF=0.5\n")          # no read covers it: an error, not a record
    locs = PE.locations_from_vcf(vcf, label=2)
    assert [(l.contig, l.pos, l.name, l.label) for l in locs] == [("chr20", 1500, "chr20:1500", 2), ("chr20", 1900, "chr20:1900", 2),
                                                                 ("chr20", 2990, "chr20:2990", 2)]
    opt = PE.EncoderOptions(window_size=100, max_reads=200)
    recs, errors = PE.encode_locations(bam, fa, locs, opt)
    assert errors == 1 and len(recs) == 2 and recs.dtype == PE.record_dtype(200, 201)
    snp, ins = recs[0], recs[1]
    assert bytes(snp["name"]).rstrip(b"\x00") == b"chr20:1500" and bytes(snp["vcfrec"]).startswith(b"chr20\t1500\t.\t")
    n = int(snp["num_reads"])
    covering = [r for r in reads if r[0] <= snp_pos - 1 < r[0] + 150]
    assert 0 < n <= 200
    # the candidate's column is image column 100: reference base there, and the reads split between ref and alt tokens
    tok = PE.BASE_ENUM
    assert snp["ref_bases"][100] == tok[ref[snp_pos - 1]] and snp["ref_bases"][99] == tok[ref[snp_pos - 2]]
    col = snp["single_reads"][:n, 100]
    assert set(np.unique(col[col > 0])) <= {tok[ref[snp_pos - 1]], tok[alt], PE.START, PE.END}
    assert (col == tok[alt]).sum() == sum(1 for r in covering if "frag" in r[1] and int(r[1][4:]) % 2 == 0)
    # no insertion anywhere near the SNP: reference line is the plain reference
    assert snp["ref_bases"].tolist() == [tok[c] for c in ref[snp_pos - 1 - 100:snp_pos + 100]]
    # strands: forward reads 2, reverse reads 1, nothing else where a token sits
    st = snp["strand"][:n]
    assert set(np.unique(st[snp["single_reads"][:n] > 0])) <= {1, 2}
    assert np.array_equal(snp["q-scores"][:n] > 0, snp["single_reads"][:n] > 0)
    # the insertion: four columns behind the candidate's column hold GGTT in the reads that carry it, 'noinsert' in the others
    m = int(ins["num_reads"])
    block = ins["single_reads"][:m, 101:105]
    with_ins = (block == [tok["G"], tok["G"], tok["T"], tok["T"]]).all(axis=1)
    without = (block == PE.NOINSERT).all(axis=1)
    present = ins["single_reads"][:m, 100] > 0
    assert with_ins.sum() > 5 and without.sum() > 5 and ((with_ins | without) | ~present).all()
    assert (ins["ref_bases"][101:105] == tok[""]).all()                   # no reference base under an insertion block
    # the records go through the dataset layer like any candidate record
    from dl4vc_amd.dataset import assemble_site
    site = assemble_site(snp, 100, np.random.RandomState(0))
    assert site.reads.shape == (100, 201) or site.reads.shape == (201, 100)


def test_converter_cli_chunks_appends_and_worker_processes(tmp_path):
    """tools/convert_bam_single_reads.py with the reference's flags: chunked appends (--locations-process-step) and worker
    processes give the same records, in input order, as one in-process pass; labels follow the VCF each location came from."""
    import subprocess
    import sys
    from dl4vc_amd import hdf5io
    rng = np.random.default_rng(21)
    ref = "".join(rng.choice(list("ACGT"), 6000))
    fa = str(tmp_path / "ref.fa")
    open(fa, "w").write(">20\n" + "\n".join(ref[i:i + 60] for i in range(0, 6000, 60)) + "\n")
    bam = str(tmp_path / "reads.bam")
    starts = np.sort(rng.integers(0, 5800, 900))
    with BamWriter(bam, [("20", 6000)]) as w:
        for i, s in enumerate(starts):
            n = int(min(150, 6000 - s))
            seq = list(ref[s:s + n])
            for j in range(n):
                if rng.random() < 0.01:
                    seq[j] = "ACGT"[int(rng.integers(0, 4))]
            w.write(0, int(s), "f%d" % i, FREVERSE if i % 3 == 0 else 0, 60, [(CMATCH, n)], "".join(seq), rng.integers(10, 41, n).tolist())
    positions = sorted(int(p) for p in rng.choice(np.arange(300, 5600), 60, replace=False))
    fp, tp = str(tmp_path / "fp.vcf"), str(tmp_path / "tp.vcf")
    head = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n"
    open(fp, "w").write(head + "".join("20\t%d\t.\t%s\tT\t50\t.\tDP=30\tGT\t0/1\n" % (p, ref[p - 1]) for p in positions[:50]))
    open(tp, "w").write(head + "".join("20\t%d\t.\t%s\tG\t50\t.\tDP=30\tGT\t1/1\n" % (p, ref[p - 1]) for p in positions[50:]))
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "convert_bam_single_reads.py")
    common = ["--input", bam, "--fasta-input", fa, "--fp_vcf", fp, "--tp_vcf", tp, "--tp_full_vcf", tp, "--max-reads", "200",
              "--max-insert-length", "10", "--max-insert-length-variant", "50", "--save-q-scores", "--save-strand"]
    one, many = str(tmp_path / "one.hdf"), str(tmp_path / "many.hdf")
    r = subprocess.run([sys.executable, tool, "--output", one, "--num-processes", "1"] + common, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Parsing errors in 0 / 60 locations" in r.stdout
    r = subprocess.run([sys.executable, tool, "--output", many, "--num-processes", "2", "--locations-process-step", "25"] + common,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    with hdf5io.CandidateFile(one) as a, hdf5io.CandidateFile(many) as b:
        assert len(a) == len(b) == 60
        ra, rb = a.read(0, 60), b.read(0, 60)
    assert ra.tobytes() == rb.tobytes()
    names = [bytes(x).rstrip(b"\x00").decode() for x in ra["name"]]
    assert names == ["20:%d" % p for p in positions[50:]] + ["20:%d" % p for p in positions[:50]]      # tp first, then fp (main():560-565)
    assert ra["label"].ravel().tolist() == [0] * 10 + [2] * 50
    assert bytes(ra["vcfrec"][0]).rstrip(b"\x00").endswith(b"\tGT:1/1") and not bytes(ra["vcfrec"][20]).rstrip(b"\x00").endswith(b"GT:0/1\tGT:0/1")
    r = subprocess.run([sys.executable, tool, "--output", one, "--num-processes", "1", "--restrict_locations"] + common, capture_output=True, text=True)
    assert r.returncode != 0 and "not supported" in r.stderr


def test_window_reader_equals_fetch_for_forward_backward_and_far_queries(tmp_path):
    rng = np.random.default_rng(8)
    p = str(tmp_path / "w.bam")
    starts = np.sort(rng.integers(0, 300000, 3000))
    with BamWriter(p, [("a", 400000), ("b", 1000)]) as w:
        for i, s in enumerate(starts):
            n = int(rng.integers(40, 151))
            w.write(0, int(s), "q%d" % i, 0, 60, [(CMATCH, n)], "A" * n, [30] * n)
        w.write(1, 10, "onb", 0, 60, [(CMATCH, 50)], "C" * 50, [30] * 50)
    for indexed in (False, True):
        if indexed:
            build_bai(p, p + ".bai")
        with BamFile(p) as bam:
            win = bamio.WindowReader(bam)
            queries = [(0, 1000, 1205), (0, 1100, 1305), (0, 1100, 1305), (0, 1300, 1505), (0, 50000, 50205), (0, 50100, 50305),
                       (0, 200, 405), (0, 299000, 299900), (0, 390000, 390100), (1, 0, 100), (0, 150000, 150205), (0, 150001, 150206)]
            for tid, lo, hi in queries:
                got = [r.name for r in win.reads(tid, lo, hi)]
                assert got == [r.name for r in bam.fetch(tid, lo, hi)], (indexed, tid, lo, hi)


def test_read_by_read_builder_equals_the_pinned_column_builder():
    """``process_tracks`` (numpy per read) against ``process_columns`` (the column-by-column form the golden vectors pin) on
    simulated pileups with insertions beyond both caps, deletions, soft clips and both strands; it declines (NotImplemented)
    what only the column order of operations reproduces: duplicated read keys, reference skips, --min-base-quality."""
    from oracle.gen_golden_pileup import simulate_reads
    from dl4vc_amd.pileup import resolve_reads
    n_fast = 0
    for seed in range(40):
        w = [100, 100, 30, 16][seed % 4]
        opt = PE.EncoderOptions(window_size=w, max_reads=200, max_insert_length=[10, 3, 0][seed % 3], max_insert_length_variant=[50, 5, 0][seed % 3])
        ref, center, reads = simulate_reads(100 + seed, w, [8, 40, 90, 300][seed % 4], duplicate_ids=(seed % 10 == 9))
        s0, stop = max(center - (w + 2), 0), center + (w + 2) + 1
        tracks = resolve_reads(reads)
        fast = PE.process_tracks(tracks, s0, stop, center, opt, ref, 0)
        cols = [PE.ColumnInput(c.reference_pos, c.query_sequences(), c.query_qualities(), c.query_ids(), ref[c.reference_pos:c.reference_pos + 1])
                for c in pileup_columns((), s0, stop, tracks=tracks)]
        slow = PE.process_columns(cols, center, opt)
        assert fast is not NotImplemented                            # (duplicated NAMES with different sequences are distinct keys)
        if slow is None:
            assert fast is None
            continue
        n_fast += 1
        for k, name in ((0, "image"), (3, "quality"), (4, "strand")):
            assert fast[k].shape == slow[k].shape and np.array_equal(fast[k], slow[k]), (seed, name)
        assert fast[1] == slow[1] and fast[2] == slow[2], seed
    assert n_fast >= 30
    ref, center, reads = simulate_reads(7, 30, 20)
    twin = reads[3]
    reads = sorted(reads + [bamio.BamRecord(0, twin.pos + 9, twin.mapq, twin.flag, twin.name, twin.cigar, twin.seq, twin.qual)], key=lambda r: r.pos)
    assert PE.process_tracks(resolve_reads(reads), max(center - 32, 0), center + 33, center, PE.EncoderOptions(window_size=30), ref, 0) is NotImplemented
    recs = [bamio.parse_record(bamio.pack_record(0, 5, "s", 0, 30, [(CMATCH, 4), (CREF_SKIP, 10), (CMATCH, 4)], "ACGTACGT", [30] * 8)[4:])]
    assert PE.process_tracks(resolve_reads(recs), 0, 40, 8, PE.EncoderOptions(window_size=16), "A" * 40, 0) is NotImplemented
    ok = [bamio.parse_record(bamio.pack_record(0, 5, "s", 0, 30, [(CMATCH, 8)], "ACGTACGT", [30] * 8)[4:])]
    assert PE.process_tracks(resolve_reads(ok), 0, 40, 8, PE.EncoderOptions(window_size=16, min_base_quality=5), "A" * 40, 0) is NotImplemented
    assert PE.process_tracks(resolve_reads(ok), 0, 40, 30, PE.EncoderOptions(window_size=16), "A" * 40, 0) is None      # POS not covered


def test_bgzf_layer_against_pythons_gzip(tmp_path):
    """BGZF is a series of gzip members: Python's own gzip module (an independent inflater + CRC check) must read what
    ``BamWriter`` wrote, byte for byte what ``BgzfReader`` returns, and the file must end with the 28-byte EOF marker block."""
    import gzip as gz
    p = str(tmp_path / "g.bam")
    rng = np.random.default_rng(4)
    with BamWriter(p, [("c", 100000)]) as w:
        for i, s in enumerate(np.sort(rng.integers(0, 90000, 2500))):
            w.write(0, int(s), "r%d" % i, 0, 30, [(CMATCH, 100)], "".join(rng.choice(list("ACGT"), 100)), rng.integers(0, 42, 100).tolist())
    raw = open(p, "rb").read()
    assert raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    plain = gz.open(p, "rb").read()
    assert plain[:4] == b"BAM\x01"
    r = bamio.BgzfReader(p)
    mine = r.read(len(plain) + 10)
    r.close()
    assert mine == plain
    # and the records parsed from gzip's bytes are the ones BamFile yields
    l_text = struct.unpack_from("<i", plain, 4)[0]
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", plain, o)[0]
    o += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", plain, o)[0]
        o += 4 + l_name + 4
    names = []
    while o < len(plain):
        size = struct.unpack_from("<i", plain, o)[0]
        names.append(bamio.parse_record(plain[o + 4:o + 4 + size]).name)
        o += 4 + size
    with BamFile(p) as bam:
        assert [x.name for x in bam] == names and len(names) == 2500


def _naive_resolution(cigar):
    """Base-by-base walk of a CIGAR, written independently of ReadTrack: per reference offset (qpos, is_del, is_skip) and the
    indel reported on the last column before an insertion / deletion run."""
    cols = []                        # [qpos, is_del, is_skip, indel]
    y = 0
    expanded = [op for op, l in cigar for _ in range(l)]
    i = 0
    while i < len(expanded):
        op = expanded[i]
        if op in (CMATCH, 7, 8):
            cols.append([y, False, False, 0])
            y += 1
        elif op in (CDEL, CREF_SKIP):
            cols.append([y, True, op == CREF_SKIP, 0])
        elif op in (CINS, CSOFT_CLIP):
            y += 1
        i += 1
    # indels: look at what follows each reference-consuming run in the op list
    ref_ops = (CMATCH, 7, 8, CDEL, CREF_SKIP)
    x = 0
    for k, (op, l) in enumerate(cigar):
        if op in ref_ops:
            x += l
            if k + 1 < len(cigar):
                nxt = [o for o, _ in cigar[k + 1:]]
                lens = [n for _, n in cigar[k + 1:]]
                v = 0
                if nxt[0] == CDEL and op != CDEL:
                    j = 0
                    while j < len(nxt) and nxt[j] == CDEL:
                        v -= lens[j]
                        j += 1
                elif nxt[0] == CINS:
                    j = 0
                    while j < len(nxt) and nxt[j] in (CINS, CPAD):
                        v += lens[j] if nxt[j] == CINS else 0
                        j += 1
                elif nxt[0] == CPAD and len(nxt) > 1:
                    j = 1
                    while j < len(nxt) and nxt[j] not in ref_ops:
                        v += lens[j] if nxt[j] == CINS else 0
                        j += 1
                cols[x - 1][3] = v
    return cols


def test_cigar_resolution_against_a_naive_walk():
    from dl4vc_amd.pileup import ReadTrack
    rng = np.random.default_rng(77)
    ops_mid = [CMATCH, CINS, CDEL, CREF_SKIP, CPAD, 7, 8]
    for trial in range(400):
        cigar = []
        if rng.random() < 0.3:
            cigar.append((CHARD_CLIP, int(rng.integers(1, 5))))
        if rng.random() < 0.4:
            cigar.append((CSOFT_CLIP, int(rng.integers(1, 9))))
        cigar.append((CMATCH, int(rng.integers(1, 12))))
        for _ in range(int(rng.integers(0, 7))):
            op = int(rng.choice(ops_mid))
            if cigar[-1][0] == op:
                continue
            cigar.append((op, int(rng.integers(1, 6))))
        if cigar[-1][0] not in (CMATCH, 7, 8):
            cigar.append((CMATCH, int(rng.integers(1, 9))))
        if rng.random() < 0.4:
            cigar.append((CSOFT_CLIP, int(rng.integers(1, 6))))
        qlen = sum(l for op, l in cigar if op in (CMATCH, CINS, CSOFT_CLIP, 7, 8))
        rec = bamio.BamRecord(0, 100, 30, 0, "t", tuple(cigar), "A" * qlen, np.full(qlen, 30, np.uint8))
        t = ReadTrack(rec)
        want = _naive_resolution(cigar)
        assert t.end - t.start == len(want) == rec.reference_end - rec.pos, cigar
        got = [[int(t.qpos[i]), bool(t.is_del[i]), bool(t.is_refskip[i]), int(t.indel[i])] for i in range(len(want))]
        assert got == want, (cigar, got, want)
