#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- generates tests/golden/pileup_encoder.json.gz: golden vectors for the image logic of the pileup
encoder (dl4vc_amd/pileup_encoder.py), computed by the REFERENCE's own helper functions.

``tools/convert_bam_single_reads.py`` cannot be imported here (it imports pysam, h5py and tqdm at module level; none of its
helpers below touch them), so this script reads its source, keeps -- by name, through ``ast`` -- the token tables and the pure
functions

    decode_base_detail, decode_query_sequences, resize_alignment_image, add_bases_to_alignment_image,
    handle_ended_sequences, center_image_on_column, trim_empty_rows, center_image_on_row_window

and executes exactly those definitions.  The per-column loop of ``process_location`` (:906-1118) and the crop / pad step of
``process_locations_chunk`` (:720-838) take pysam objects and are transcribed below around those functions (``drive_columns``,
``crop_and_pad``): every array operation on the images is the reference's own code, the bookkeeping between the calls is a
transcription.  Inputs are synthetic pileup columns (read strings, qualities, read ids, reference bases) of simulated reads
with substitutions, insertions, deletions, soft clips, both strands and duplicated read ids; they are stored in the fixture,
so the test needs neither this script nor the reference.

Run from the repository root (needs /root/reference):  python oracle/gen_golden_pileup.py
"""
import ast
import gzip
import hashlib
import json
import os
import re
import sys
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REFERENCE = "/root/reference/tools/convert_bam_single_reads.py"
KEEP_FUNCS = {"decode_base_detail", "decode_query_sequences", "resize_alignment_image", "add_bases_to_alignment_image",
              "handle_ended_sequences", "center_image_on_column", "trim_empty_rows", "center_image_on_row_window"}
KEEP_NAMES = {"debug", "base_enum", "real_bases_set", "enum_base", "STRAND_PAD", "STRAND_LOWER", "STRAND_UPPER", "strand_enum"}


def reference_namespace():
    tree = ast.parse(open(REFERENCE).read())
    body = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in KEEP_FUNCS:
            body.append(node)
        elif isinstance(node, ast.Assign) and all(isinstance(t, ast.Name) and t.id in KEEP_NAMES for t in node.targets):
            body.append(node)
    ns = {"np": np, "re": re}
    exec(compile(ast.Module(body=body, type_ignores=[]), REFERENCE, "exec"), ns)
    missing = (KEEP_FUNCS | KEEP_NAMES) - set(ns)
    assert not missing, missing
    return SimpleNamespace(**ns)


def drive_columns(R, columns, center_position, window_size, max_insert_length, max_insert_length_variant):
    """process_location (:846-1118) with the pysam calls replaced by the prepared columns."""
    args = SimpleNamespace(save_q_scores=True, save_strand=True, debug=False)
    MAX_INSERT_LENGTH = max_insert_length
    MAX_INSERT_VARIANT = max(max_insert_length_variant, MAX_INSERT_LENGTH)
    window_size += 2
    max_reads, read_window = 1200, 3 * window_size
    alignment_image = np.full((max_reads, read_window), R.base_enum['pad'], dtype=np.uint8)
    read_quality_image = np.full(alignment_image.shape, 0, dtype=np.uint8)
    strand_image = np.full(alignment_image.shape, 0, dtype=np.uint8)
    MAX = 1000
    idx, prev_col_offset, col_offset, local_max_col_offset = 0, 0, 1, 0
    read_row_dict, reads_offset, col_reference_map = {}, 0, {}
    for c in columns:
        if idx > MAX:
            break
        (query_seq_bases, query_seq_inserts) = R.decode_query_sequences(c["sequences"])
        assert len(query_seq_bases) == len(c["ids"]), "mismatch between pileups & sequences"
        query_qualities = c["qualities"]
        query_seq_ids = c["ids"]
        min_row_size = len(read_row_dict) + reads_offset + len(query_seq_ids) + 1
        min_col_size = col_offset + MAX_INSERT_VARIANT + max(10, MAX_INSERT_VARIANT)
        alignment_image = R.resize_alignment_image(min_row_size, min_col_size, alignment_image)
        read_quality_image = R.resize_alignment_image(min_row_size, min_col_size, read_quality_image)
        strand_image = R.resize_alignment_image(min_row_size, min_col_size, strand_image)
        for i, name in enumerate(query_seq_ids):
            if not (name in read_row_dict.keys()):
                read_row_dict[name] = len(read_row_dict) + reads_offset
        is_variant = (c["reference_pos"] == center_position - 1)
        local_max_col_offset = R.add_bases_to_alignment_image(
            query_seq_bases, query_seq_inserts, query_seq_ids, alignment_image, col_offset, prev_col_offset, read_row_dict,
            query_qualities=query_qualities, read_quality_image=read_quality_image, strand_image=strand_image,
            MAX_INSERT_LENGTH=(MAX_INSERT_VARIANT if is_variant else MAX_INSERT_LENGTH), args=args)
        (read_row_dict, reads_offset, finished_rows) = R.handle_ended_sequences(
            query_seq_bases, query_seq_ids, alignment_image, read_row_dict, reads_offset, col_offset, local_max_col_offset,
            query_qualities=query_qualities, read_quality_image=read_quality_image, strand_image=strand_image, args=args)
        col_reference_map[idx] = (col_offset, c["reference_pos"], c["ref_base"])
        idx += 1
        prev_col_offset = col_offset
        col_offset = col_offset + 1 + local_max_col_offset
        local_max_col_offset = 0
    for row_num in range(strand_image.shape[0]):
        num_pads = np.sum(strand_image[row_num, :] == R.STRAND_PAD)
        if num_pads == 0:
            continue
        row_nopad = strand_image[row_num, :] * ~(strand_image[row_num, :] == R.STRAND_PAD)
        strand_value = max(row_nopad)
        if strand_value == 0:
            strand_value = R.STRAND_UPPER
        strand_image[row_num, :][strand_image[row_num, :] == R.STRAND_PAD] = strand_value
    n = len(read_row_dict) + reads_offset
    image_sample = alignment_image[:n, :col_offset + 1]
    center_index = -1
    for i in col_reference_map:
        if int(col_reference_map[i][1]) == center_position - 1:
            center_index = col_reference_map[i][0]
            break
    if center_index == -1:
        return None
    return image_sample, center_index, col_reference_map, read_quality_image[:n, :col_offset + 1], strand_image[:n, :col_offset + 1]


def crop_and_pad(R, result, window_size, max_reads):
    """process_locations_chunk (:720-838) for one image; None where the reference counts an error."""
    single_read, center_index, reference_index, quality_read, strand_read = result
    TOTAL_SINGLE_READS, TOTAL_COLUMNS = max_reads, 2 * window_size + 1
    reference_bases = np.full(single_read.shape[1], R.base_enum[''], dtype=np.uint8)
    for k in reference_index.keys():
        off, ref_pos, ref_base = reference_index[k]
        reference_bases[off] = R.base_enum[ref_base]
    (min_col_idx, max_col_idx) = R.center_image_on_column(single_read, center_index, window_size)
    single_read = single_read[:, min_col_idx:max_col_idx]
    quality_read = quality_read[:, min_col_idx:max_col_idx]
    strand_read = strand_read[:, min_col_idx:max_col_idx]
    single_read = R.trim_empty_rows(single_read, "top")
    quality_read = R.trim_empty_rows(quality_read, "top")
    strand_read = R.trim_empty_rows(strand_read, "top")
    (min_read, max_read) = R.center_image_on_row_window(single_read, TOTAL_SINGLE_READS)
    single_read = single_read[min_read:max_read, :]
    num_reads = single_read.shape[0]
    quality_read = quality_read[min_read:max_read, :]
    strand_read = strand_read[min_read:max_read, :]
    if quality_read.shape != single_read.shape or strand_read.shape != single_read.shape or num_reads <= 0:
        return None
    reference_bases = reference_bases[min_col_idx:max_col_idx]
    single_read_pad = np.zeros((TOTAL_SINGLE_READS, TOTAL_COLUMNS), dtype=np.uint8)
    reference_bases_pad = np.zeros((TOTAL_COLUMNS), dtype=np.uint8)
    idx_offset = (window_size) - (center_index - min_col_idx)
    rows = min(TOTAL_SINGLE_READS, single_read.shape[0])
    single_read_pad[:rows, idx_offset:idx_offset + single_read.shape[1]] = single_read
    reference_bases_pad[idx_offset:idx_offset + single_read.shape[1]] = reference_bases
    quality_read_pad = np.zeros((TOTAL_SINGLE_READS, TOTAL_COLUMNS), dtype=np.uint8)
    quality_read_pad[:rows, idx_offset:idx_offset + single_read.shape[1]] = quality_read
    strand_read_pad = np.zeros((TOTAL_SINGLE_READS, TOTAL_COLUMNS), dtype=np.uint8)
    strand_read_pad[:rows, idx_offset:idx_offset + single_read.shape[1]] = strand_read
    return single_read_pad, reference_bases_pad, min(num_reads, TOTAL_SINGLE_READS), quality_read_pad, strand_read_pad


# ------------------------------------------------------------------------------------------------------
# synthetic pileups
# ------------------------------------------------------------------------------------------------------
def simulate_reads(seed, window_size, n_reads, contig_len=600, duplicate_ids=False):
    """(reference sequence, candidate POS, simulated reads in coordinate order)."""
    from dl4vc_amd.bamio import BamRecord, CMATCH, CINS, CDEL, CSOFT_CLIP, FREVERSE
    rng = np.random.default_rng(seed)
    ref = "".join(rng.choice(list("ACGT"), contig_len))
    if seed % 3 == 0:
        ref = ref[:250] + ref[250:300].lower() + ref[300:]              # soft-masked stretch
    center = contig_len // 2 + int(rng.integers(-20, 20))                # VCF POS (1-based)
    reads = []
    for i in range(n_reads):
        length = int(rng.integers(60, 151))
        start = int(rng.integers(max(0, center - window_size - 140), center + window_size + 20))
        cigar, seq, rpos = [], [], start
        if rng.random() < 0.2:
            k = int(rng.integers(1, 12))
            cigar.append((CSOFT_CLIP, k))
            seq.extend(rng.choice(list("ACGT"), k))
        remaining = length
        while remaining > 0 and rpos < contig_len - 1:
            m = int(min(remaining, rng.integers(5, 60), contig_len - 1 - rpos))
            if m <= 0:
                break
            block = list(ref[rpos:rpos + m].upper())
            for j in range(m):
                if rng.random() < 0.03:
                    block[j] = str(rng.choice(list("ACGTN")))
            # the candidate: a share of reads carries an insertion / deletion / SNP exactly there
            cigar.append((CMATCH, m))
            seq.extend(block)
            rpos += m
            remaining -= m
            if remaining <= 0:
                break
            u = rng.random()
            near = abs(rpos - center) < 4
            if u < (0.6 if near else 0.15):
                k = int(rng.integers(1, 60 if near and rng.random() < 0.3 else 14))
                cigar.append((CINS, k))
                seq.extend(rng.choice(list("ACGT"), k))
                remaining -= k
            elif u < (0.9 if near else 0.3):
                k = int(rng.integers(1, 12))
                cigar.append((CDEL, k))
                rpos += k
        while cigar and cigar[-1][0] != CMATCH:                          # end on a match
            op, k = cigar.pop()
            if op == CINS:
                del seq[-k:]
        if not cigar:
            continue
        name = "read%d" % (i // 2 if duplicate_ids and i % 7 == 0 else i)
        qual = rng.integers(2, 42, len(seq)).astype(np.uint8)
        flag = FREVERSE if rng.random() < 0.5 else 0
        reads.append(BamRecord(0, start, int(rng.integers(0, 61)), flag, name, tuple(cigar), "".join(seq), qual))
    reads.sort(key=lambda r: r.pos)
    return ref, center, reads


def simulate_case(seed, window_size, n_reads, contig_len=600, duplicate_ids=False, deep=False):
    """Columns of simulated reads around a candidate position, made with dl4vc_amd.pileup (inputs only: the fixture stores them)."""
    from dl4vc_amd.pileup import pileup_columns
    ref, center, reads = simulate_reads(seed, window_size, n_reads, contig_len, duplicate_ids)
    w = window_size + 2
    s0, stop = max(center - w, 0), center + w + 1
    cols = []
    for c in pileup_columns(reads, s0, stop):
        cols.append({"reference_pos": c.reference_pos, "sequences": c.query_sequences(), "qualities": c.query_qualities(),
                     "ids": c.query_ids(), "ref_base": ref[c.reference_pos:c.reference_pos + 1]})
    return {"center_position": center, "columns": cols}


def digest(a):
    """Large expected arrays are stored as shape + SHA-256 of their uint8 bytes; small ones in full."""
    a = np.ascontiguousarray(a, np.uint8)
    d = {"shape": list(a.shape), "sha256": hashlib.sha256(a.tobytes()).hexdigest()}
    if a.size <= 4000:
        d["values"] = a.tolist()
    return d


def compact(case):
    """Read ids are ``name:sequence``: store each once (``id_table``) and the columns' ids as indices into it."""
    table = {}
    for col in case["columns"]:
        col["ids"] = [table.setdefault(i, len(table)) for i in col["ids"]]
    case["id_table"] = list(table)
    return case


def main():
    R = reference_namespace()
    out = {"decode": [], "cases": []}
    # decode_base_detail on the read strings the encoder can meet
    strings = ["A", "a", "t", "G", "c", "N", "n", "*", "^!A", "^~c", "^]T$", "g$", "A+1C", "a+3tcg", "C+12ACGTACGTACGT", "t-2nn",
               "G-11NNNNNNNNNNN", "*+2AC", "^IA+2GG$", "*-3NNN", "^", "^5", "^5$"]
    for s in strings:
        try:
            out["decode"].append({"s": s, "want": [int(v) if not isinstance(v, list) else [int(x) for x in v] for v in R.decode_base_detail(s)]})
        except Exception as e:                                          # noqa: BLE001
            out["decode"].append({"s": s, "raises": type(e).__name__})
    specs = [(1, 20, 12, 200, 10, 50, False), (2, 100, 40, 200, 10, 50, False), (3, 100, 90, 200, 10, 50, True),
             (4, 100, 260, 200, 10, 50, False), (5, 30, 25, 10, 3, 5, False), (6, 100, 30, 200, 0, 0, False),
             (7, 100, 60, 50, 10, 50, True), (8, 16, 6, 200, 10, 50, False), (9, 100, 1300, 1000, 10, 50, False)]
    for seed, w, n_reads, max_reads, mil, milv, dup in specs:
        case = simulate_case(seed, w, n_reads, duplicate_ids=dup)
        case.update(window_size=w, max_reads=max_reads, max_insert_length=mil, max_insert_length_variant=milv)
        res = drive_columns(R, case["columns"], case["center_position"], w, mil, milv)
        if res is None:
            case["want"] = None
        else:
            img, center, colmap, q, s = res
            rec = crop_and_pad(R, res, w, max_reads)
            case["want"] = {"image": digest(img), "center": int(center), "quality": digest(q), "strand": digest(s),
                            "colmap": [[int(k), int(v[0]), int(v[1]), v[2]] for k, v in colmap.items()],
                            "record": None if rec is None else {"single_reads": digest(rec[0]), "ref_bases": digest(rec[1]),
                                                                "num_reads": int(rec[2]), "q-scores": digest(rec[3]),
                                                                "strand": digest(rec[4])}}
        out["cases"].append(compact(case))
        print("case seed %d: %d columns, image %s, record %s" % (seed, len(case["columns"]),
              None if res is None else res[0].shape, None if case["want"] is None or case["want"]["record"] is None else "ok"))
    # a location whose own position has no column
    c = simulate_case(11, 20, 3)
    c["columns"] = [col for col in c["columns"] if col["reference_pos"] != c["center_position"] - 1]
    c.update(window_size=20, max_reads=50, max_insert_length=10, max_insert_length_variant=50)
    assert drive_columns(R, c["columns"], c["center_position"], 20, 10, 50) is None
    c["want"] = None
    out["cases"].append(compact(c))
    path = os.path.join(ROOT, "tests", "golden", "pileup_encoder.json.gz")
    with gzip.open(path, "wt") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
