"""Pileup encoder: BAM + candidate VCF -> the candidate records main.py scores.

Restates ``tools/convert_bam_single_reads.py``: the per-location image builder ``process_location`` (:846-1118) with its
helpers ``decode_base_detail`` (:84-147), ``decode_query_sequences`` (:198-224), ``resize_alignment_image`` (:226-254),
``add_bases_to_alignment_image`` (:256-345), ``handle_ended_sequences`` (:347-398), the crop / centre / pad step of
``process_locations_chunk`` (:676-845: ``center_image_on_column`` :400-429, ``trim_empty_rows`` :431-461,
``center_image_on_row_window`` :463-478) and the record layout it saves (:703-708), plus ``get_locations_from_vcf``
(:160-196) on a plain-text VCF.

How a location becomes a record, as the reference does it:

* pileup columns of ``[POS - w - 2, POS + w + 3)`` (0-based, ``w`` = ``--window-size``), one image column per reference
  position plus, behind it, as many columns as the longest insertion any read carries there (capped at
  ``--max-insert-length``, ``--max-insert-length-variant`` on the candidate's own column); reads without that insertion
  get ``noinsert`` (8) in those columns;
* one image row per read, keyed ``name:sequence``, rows in order of first appearance and never re-used; ``start`` (6) in the
  column before a read's first base, ``end`` (7) in the column after its last one (after that column's insertion block);
* three planes: base tokens, base qualities, strand (1 lower / reverse, 2 upper / forward; deletions take the read's strand);
* crop to ``w`` columns either side of the candidate's column, drop leading all-zero rows, keep the middle ``--max-reads``
  rows, pad into ``(max_reads, 2w + 1)``.

Parity: the image logic above is PINNED -- tests/golden/pileup_encoder.json.gz holds outputs of the reference's own helper
functions (executed from its source by oracle/gen_golden_pileup.py on synthetic pileup columns) and of this module's
``process_columns`` glue is compared against a transcription of the reference's loop driven by those helpers.  The BAM side
(dl4vc_amd/bamio.py, dl4vc_amd/pileup.py) is UNPINNED: no htslib / pysam in the image.
Differences kept on purpose: records come out in input order (the reference's ``imap_unordered`` order is arbitrary);
IUPAC ambiguity codes in a read take their strand from their case instead of raising ``KeyError`` (the reference's strand
table has no entry for them, :47-56).
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .hdf5_schema import record_dtype

# token tables (tools/convert_bam_single_reads.py:38-56)
BASE_ENUM = {"A": 1, "a": 1, "T": 2, "t": 2, "U": 2, "u": 2, "G": 3, "g": 3, "C": 4, "c": 4,
             "": 5, "-": 5, "*": 5, "N": 5, "n": 5, "X": 5, "x": 5, ".": 5, ",": 5,
             "start": 6, "e": 7, "end": 7, "noinsert": 8, "pad": 0,
             "unk": 9, "?": 9, "M": 9, "m": 9, "K": 9, "k": 9, "R": 9, "r": 9, "Y": 9, "y": 9,
             "S": 9, "s": 9, "W": 9, "w": 9, "B": 9, "b": 9, "V": 9, "v": 9, "H": 9, "h": 9, "D": 9, "d": 9}
PAD, START, END, NOINSERT = 0, 6, 7, 8
STRAND_PAD, STRAND_LOWER, STRAND_UPPER = 200, 1, 2
_STRAND = {"A": 2, "a": 1, "T": 2, "t": 1, "G": 2, "g": 1, "C": 2, "c": 1, "": 200, "-": 200, "*": 200, "N": 2, "n": 1,
           "M": 2, "m": 1, "?": 200}
_INDEL = re.compile(r"(\d+)(\D+)")


@dataclass
class EncoderOptions:
    """The converter's flags that shape a record (argparse defaults at :481-527; call_variants.sh:87-98 passes
    ``--max-reads 200 --max-insert-length 10 --max-insert-length-variant 50 --save-q-scores --save-strand``)."""
    window_size: int = 100
    max_reads: int = 1000
    max_insert_length: int = 10
    max_insert_length_variant: int = 50
    min_base_quality: int = 0


def decode_base(base_str: str):
    """(start, base token, deletion length, end, strand, insert tokens) of one read string (:84-147)."""
    start = end = base = deletion = 0
    strand = STRAND_PAD
    insert: List[int] = []
    if base_str[0] == "^":
        if len(base_str) > 1:
            start = ord(base_str[1]) - 10
        else:
            return 1, base, deletion, end, strand, insert
        if len(base_str) > 2:
            base_str = base_str[2:]
        else:
            return start, base, deletion, end, strand, insert
    if base_str[-1] == "$":
        end = 1
        base_str = base_str[:-1]
    c = base_str[0]
    base = BASE_ENUM[c]
    strand = _STRAND[c] if c in _STRAND else (STRAND_UPPER if c.isupper() else STRAND_LOWER if c.islower() else STRAND_PAD)
    if len(base_str) > 1:
        m = _INDEL.match(base_str[2:])
        if base_str[1] == "+":
            insert = [BASE_ENUM[b] for b in m.group(2)]
            assert int(m.group(1)) == len(insert)
        elif base_str[1] == "-":
            deletion = int(m.group(1))
        else:
            raise AssertionError("unparsable indel in |%s|" % base_str)
    return start, base, deletion, end, strand, insert


def _grow(img: np.ndarray, rows: int, cols: int) -> np.ndarray:
    """:226-254: double (at least to the size asked) whichever dimension is too small; new cells are ``pad``."""
    if rows > img.shape[0]:
        new = np.zeros((max(img.shape[0] * 2, rows), img.shape[1]), np.uint8)
        new[:img.shape[0]] = img
        img = new
    if cols > img.shape[1]:
        new = np.zeros((img.shape[0], max(img.shape[1] * 2, cols)), np.uint8)
        new[:, :img.shape[1]] = img
        img = new
    return img


@dataclass
class ColumnInput:
    """What the image builder takes from one pileup column."""
    reference_pos: int
    sequences: Sequence[str]
    qualities: Sequence[int]
    ids: Sequence[str]
    ref_base: str


def process_columns(columns: Iterable[ColumnInput], center_position: int, opt: EncoderOptions):
    """The loop of ``process_location`` (:906-1118) over the columns of one location.  ``center_position`` is the VCF POS
    (1-based).  Returns (bases, centre column, {i: (column, reference position, reference base)}, qualities, strands) or
    None when the candidate's own position has no column (the reference returns ``([], -1, [])``)."""
    window = opt.window_size + 2
    max_var = max(opt.max_insert_length_variant, opt.max_insert_length)
    img = np.zeros((1200, 3 * window), np.uint8)
    qimg = np.zeros_like(img)
    simg = np.zeros_like(img)
    rows: Dict[str, int] = {}
    finished = 0
    idx, prev_col, col = 0, 0, 1
    colmap: Dict[int, Tuple[int, int, str]] = {}
    for c in columns:
        if idx > 1000:
            break
        details = [decode_base(s) for s in c.sequences]
        keep = [i for i, d in enumerate(details) if d[1] != 0]               # (:212-214)
        assert len(keep) == len(c.ids), "mismatch between pileups & sequences"
        details = [details[i] for i in keep]
        quals, ids = c.qualities, c.ids
        need_rows = len(rows) + finished + len(ids) + 1
        need_cols = col + max_var + max(10, max_var)
        img, qimg, simg = _grow(img, need_rows, need_cols), _grow(qimg, need_rows, need_cols), _grow(simg, need_rows, need_cols)
        for name in ids:
            if name not in rows:
                rows[name] = len(rows) + finished
        cap = max_var if c.reference_pos == center_position - 1 else opt.max_insert_length
        # ---- add_bases_to_alignment_image (:256-345)
        r = [rows[name] for name in ids]
        for i, d in enumerate(details):
            img[r[i], col], qimg[r[i], col], simg[r[i], col] = d[1], quals[i], d[4]
        for i, d in enumerate(details):
            if d[0]:
                img[r[i], prev_col], qimg[r[i], prev_col], simg[r[i], prev_col] = START, quals[i], d[4]
        longest = 0
        if cap > 0:
            for i, d in enumerate(details):
                ins = d[5][:cap]
                if ins:
                    img[r[i], col + 1:col + 1 + len(ins)] = ins
                    qimg[r[i], col + 1:col + 1 + len(ins)] = quals[i]
                    simg[r[i], col + 1:col + 1 + len(ins)] = d[4]
                    longest = max(longest, len(ins))
        if longest:
            for ri in r:
                blk = img[ri, col + 1:col + 1 + longest]
                blk[blk == PAD] = NOINSERT
        # ---- handle_ended_sequences (:347-398)
        done = [i for i, d in enumerate(details) if d[3]]
        for i in done:
            e = col + longest + 1
            img[r[i], e], qimg[r[i], e], simg[r[i], e] = END, quals[i], details[i][4]
        for i in set(done):
            rows.pop(ids[i], None)
            finished += 1
        colmap[idx] = (col, c.reference_pos, c.ref_base)
        idx += 1
        prev_col = col
        col = col + 1 + longest
    # deletions ('*') carry no strand: every row takes its own strand, forward when it has none (:1063-1078)
    for k in range(simg.shape[0]):
        row = simg[k]
        pads = row == STRAND_PAD
        if pads.any():
            v = int(row[~pads].max()) if (~pads).any() else 0
            row[pads] = v if v else STRAND_UPPER
    n_rows = len(rows) + finished
    center = -1
    for i in colmap:
        if colmap[i][1] == center_position - 1:
            center = colmap[i][0]
            break
    if center == -1:
        return None
    return img[:n_rows, :col + 1], center, colmap, qimg[:n_rows, :col + 1], simg[:n_rows, :col + 1]


# ---- the same images built read by read (numpy per row instead of Python per pileup entry) --------------------------------
_TOKEN = np.zeros(256, np.uint8)
_KNOWN = np.zeros(256, bool)
for _c, _v in BASE_ENUM.items():
    if len(_c) == 1:
        _TOKEN[ord(_c)], _KNOWN[ord(_c)] = _v, True


def process_tracks(tracks, s0: int, stop: int, center_position: int, opt: EncoderOptions, ref: str, ref_start: int):
    """``process_columns`` for the common case, one read at a time: same result, ~10x faster.  Returns ``NotImplemented`` when
    the location needs the column-by-column path -- two reads sharing a ``name:sequence`` key (they share and swap image
    rows in the reference), a reference skip, a base outside the token table, more than 1000 columns, a depth beyond
    pysam's cap or ``--min-base-quality`` -- so those keep the reference's exact column order of operations."""
    if opt.min_base_quality > 0 or len(tracks) > 8000:
        return NotImplemented
    tracks = [t for t in tracks if t.end > s0 and t.start < stop]
    keys = {t.key for t in tracks}
    if len(keys) != len(tracks):
        return NotImplemented
    n_pos = stop - s0
    cover = np.zeros(n_pos + 1, np.int32)
    longest = np.zeros(n_pos, np.int32)
    cap = np.full(n_pos, opt.max_insert_length, np.int32)
    ci = center_position - 1 - s0
    if 0 <= ci < n_pos:
        cap[ci] = max(opt.max_insert_length_variant, opt.max_insert_length)
    spans = []
    for t in tracks:
        lo, hi = max(t.start, s0), min(t.end, stop)
        if t.is_refskip[lo - t.start:hi - t.start].any():
            return NotImplemented
        cover[lo - s0] += 1
        cover[hi - s0] -= 1
        ins = t.indel[lo - t.start:hi - t.start]
        if (ins > 0).any():
            np.maximum(longest[lo - s0:hi - s0], np.minimum(np.maximum(ins, 0), cap[lo - s0:hi - s0]), out=longest[lo - s0:hi - s0])
        spans.append((lo, hi))
    covered = np.cumsum(cover[:-1]) > 0
    pos_idx = np.nonzero(covered)[0]
    if len(pos_idx) == 0 or not (0 <= ci < n_pos and covered[ci]):
        return None
    if len(pos_idx) > 1001:
        return NotImplemented
    width = 1 + longest[pos_idx]
    col_of = np.zeros(n_pos, np.int64)
    col_of[pos_idx] = 1 + np.concatenate(([0], np.cumsum(width)[:-1]))
    prev_col = np.zeros(n_pos, np.int64)
    prev_col[pos_idx] = np.concatenate(([0], col_of[pos_idx][:-1]))
    end_col = int(col_of[pos_idx[-1]] + width[-1])                  # the column the loop would give the next position
    order = sorted(range(len(tracks)), key=lambda i: (spans[i][0], i))
    n_rows = len(tracks)
    n_cols = end_col + 1
    img = np.zeros((n_rows, n_cols), np.uint8)
    qimg = np.zeros_like(img)
    simg = np.zeros_like(img)
    for row, i in enumerate(order):
        t = tracks[i]
        rec = t.rec
        lo, hi = spans[i]
        a, b = lo - t.start, hi - t.start
        cols = col_of[lo - s0:hi - s0]
        qpos, dele = t.qpos[a:b], t.is_del[a:b]
        seq = np.frombuffer(rec.seq.encode("ascii"), np.uint8)
        inside = qpos < len(seq)
        chars = np.where(inside, seq[np.minimum(qpos, len(seq) - 1)], ord("N"))
        if not _KNOWN[chars].all():
            return NotImplemented
        strand = STRAND_LOWER if rec.is_reverse else STRAND_UPPER
        img[row, cols] = np.where(dele, BASE_ENUM["*"], _TOKEN[chars])
        quals = np.where(qpos < len(rec.qual), rec.qual[np.minimum(qpos, len(rec.qual) - 1)], 0).astype(np.uint8)
        qimg[row, cols] = quals
        simg[row, cols] = np.where(dele, STRAND_PAD, strand)
        if t.start >= s0:                                            # head column inside the window
            pc = prev_col[lo - s0]
            img[row, pc], qimg[row, pc], simg[row, pc] = START, quals[0], STRAND_PAD if dele[0] else strand
        ins = t.indel[a:b]
        lg = longest[lo - s0:hi - s0]
        for k in np.nonzero(lg > 0)[0]:
            c0, m = int(cols[k]) + 1, int(lg[k])
            st = STRAND_PAD if dele[k] else strand
            if ins[k] > 0:
                n_ins = min(int(ins[k]), int(cap[lo - s0 + k]))
                if n_ins > 0:
                    q0 = int(qpos[k])
                    letters = [rec.seq[q0 + j] if q0 + j < len(rec.seq) else "N" for j in range(1, n_ins + 1)]
                    toks = _TOKEN[np.frombuffer("".join(letters).encode("ascii"), np.uint8)]
                    if not _KNOWN[np.frombuffer("".join(letters).encode("ascii"), np.uint8)].all():
                        return NotImplemented
                    img[row, c0:c0 + n_ins] = toks
                    qimg[row, c0:c0 + n_ins] = quals[k]
                    simg[row, c0:c0 + n_ins] = st
            blk = img[row, c0:c0 + m]
            blk[blk == PAD] = NOINSERT
        if t.end <= stop:                                            # tail column inside the window
            k = b - a - 1
            e = int(cols[k]) + int(lg[k]) + 1
            img[row, e], qimg[row, e], simg[row, e] = END, quals[k], STRAND_PAD if dele[k] else strand
        pads = simg[row] == STRAND_PAD
        if pads.any():
            rest = simg[row][~pads]
            v = int(rest.max()) if len(rest) else 0
            simg[row][pads] = v if v else STRAND_UPPER
    colmap = {}
    for i, p in enumerate(pos_idx):
        o = s0 + int(p) - ref_start
        colmap[i] = (int(col_of[p]), s0 + int(p), ref[o:o + 1])
    return img, int(col_of[ci]), colmap, qimg, simg


def _trim_top(image: np.ndarray) -> np.ndarray:
    """``trim_empty_rows(image, "top")`` (:431-461): rows before the first one with a non-zero sum go (none when all are zero)."""
    sums = image.sum(axis=1)
    nz = np.nonzero(sums > 0)[0]
    return image[int(nz[0]):] if len(nz) else image


def finish_record(result, location, opt: EncoderOptions, dtype: np.dtype):
    """The per-image part of ``process_locations_chunk`` (:720-838): one structured record, or None where the reference counts
    an error (its planes disagree in shape after trimming, or no read row is left)."""
    bases, center, colmap, quals, strands = result
    w, cols = opt.window_size, 2 * opt.window_size + 1
    ref_line = np.full(bases.shape[1], BASE_ENUM[""], np.uint8)
    for off, _pos, base in colmap.values():
        ref_line[off] = BASE_ENUM[base]
    lo = max(0, center - w)
    hi = min(center + w + 1, bases.shape[1])
    bases, quals, strands = _trim_top(bases[:, lo:hi]), _trim_top(quals[:, lo:hi]), _trim_top(strands[:, lo:hi])
    n = bases.shape[0]
    first = max(0, int((n - opt.max_reads) / 2))
    last = min(first + opt.max_reads, n)
    bases, quals, strands = bases[first:last], quals[first:last], strands[first:last]
    if quals.shape != bases.shape or strands.shape != bases.shape or bases.shape[0] == 0:
        return None
    rec = np.zeros((), dtype)
    off = w - (center - lo)
    k = min(opt.max_reads, bases.shape[0])
    rec["single_reads"][:k, off:off + bases.shape[1]] = bases
    rec["q-scores"][:k, off:off + bases.shape[1]] = quals
    rec["strand"][:k, off:off + bases.shape[1]] = strands
    rec["ref_bases"][off:off + bases.shape[1]] = ref_line[lo:hi]
    rec["num_reads"] = k
    rec["name"] = location.name.encode()[:dtype["name"].itemsize]
    rec["label"] = location.label
    rec["vcfrec"] = location.vcf_string.encode()[:dtype["vcfrec"].itemsize]
    # 'ref' / 'reads' summary planes: NaN for VCF-derived locations in the reference (:195), i.e. zeros once cast
    return rec


@dataclass
class Location:
    contig: str
    pos: int                       # VCF POS (1-based)
    name: str
    label: int
    vcf_string: str


def locations_from_vcf(path: str, label: int, full_vcf: Optional[str] = None) -> List[Location]:
    """``get_locations_from_vcf`` (:160-196) on a plain-text (optionally gzip / BGZF) VCF: one location per record, named
    ``contig:pos``, carrying the record's text (+ ``\\tGT:<genotype>`` when ``full_vcf`` has a genotype for that position)."""
    def lines(p):
        import gzip
        opener = gzip.open if open(p, "rb").read(2) == b"\x1f\x8b" else open
        with opener(p, "rt") as f:
            for line in f:
                if line.strip() and not line.startswith("#"):
                    yield line.rstrip("\r\n")

    genotypes = {}
    if full_vcf:
        for line in lines(full_vcf):
            p = line.split("\t")
            if len(p) >= 10 and p[-2].split(":")[0] == "GT":
                genotypes["%s:%s" % (p[0], p[1])] = "GT:%s" % p[-1].split(":")[0]
    out = []
    for line in lines(path):
        p = line.split("\t")
        name = "%s:%s" % (p[0], p[1])
        text = line.strip()
        if name in genotypes:
            text += "\t%s" % genotypes[name]
        out.append(Location(p[0], int(p[1]), name, label, text))
    return out


def encode_location(bam, fasta, loc: Location, opt: EncoderOptions, reader=None):
    """``process_location`` (:846-1118) for one candidate against an open ``BamFile`` / ``FastaFile``.  ``reader``: a
    ``bamio.WindowReader`` on ``bam`` shared by a run of locations (each alignment is then parsed once, not once per
    location)."""
    from .pileup import pileup_columns, resolve_reads
    window = opt.window_size + 2
    start, stop = loc.pos - window, loc.pos + window + 1
    tid = bam.get_tid(loc.contig)
    if tid < 0:
        return None
    s0 = max(start, 0)
    ref = fasta.fetch(loc.contig, s0, stop + 64)

    tracks = resolve_reads(reader.reads(tid, s0, stop) if reader is not None else bam.fetch(tid, s0, stop))
    fast = process_tracks(tracks, s0, stop, loc.pos, opt, ref, s0)
    if fast is not NotImplemented:
        return fast

    def cols():
        for c in pileup_columns((), s0, stop, tracks=tracks):
            seqs = c.query_sequences(opt.min_base_quality)
            o = c.reference_pos - s0
            yield ColumnInput(c.reference_pos, seqs, c.query_qualities(opt.min_base_quality), c.query_ids(opt.min_base_quality),
                              ref[o:o + 1])
    return process_columns(cols(), loc.pos, opt)


def encode_locations(bam_path: str, fasta_path: str, locations: Sequence[Location], opt: EncoderOptions,
                     native: Optional[bool] = None, threads: int = 1) -> Tuple[np.ndarray, int]:
    """Records for ``locations`` in input order and the number of locations that produced none.

    ``native`` (default: when libdl4vc_loader.so is built): the image planes come from the C++ encoder (``pe_encode``:
    BGZF / BAM / CIGAR / image builder in ``threads`` worker threads); a location it declines (status 2: the cases
    ``process_tracks`` hands to the column-by-column builder) is encoded here, so the records are the same bytes either
    way (tests/test_pileup_native.py)."""
    from .bamio import BamFile, FastaFile, WindowReader
    from . import loader
    dtype = record_dtype(opt.max_reads, 2 * opt.window_size + 1)
    out = np.zeros(len(locations), dtype)
    n = errors = 0
    use_native = loader.available() if native is None else native
    planes = None
    if use_native and len(locations):
        with loader.NativePileupEncoder(bam_path, fasta_path, opt.window_size, opt.max_reads, opt.max_insert_length,
                                        opt.max_insert_length_variant, opt.min_base_quality) as enc:
            planes = enc.encode([l.contig for l in locations], [l.pos for l in locations], threads)
    bam = fasta = reader = None
    try:
        for i, loc in enumerate(locations):
            status = int(planes[5][i]) if planes is not None else 2
            if status == 1:
                rec = np.zeros((), dtype)
                rec["single_reads"], rec["q-scores"], rec["strand"] = planes[0][i], planes[1][i], planes[2][i]
                rec["ref_bases"], rec["num_reads"] = planes[3][i], planes[4][i]
                rec["name"] = loc.name.encode()[:dtype["name"].itemsize]
                rec["label"] = loc.label
                rec["vcfrec"] = loc.vcf_string.encode()[:dtype["vcfrec"].itemsize]
            elif status == 0:
                rec = None
            else:
                if bam is None:
                    bam, fasta = BamFile(bam_path), FastaFile(fasta_path)
                    reader = WindowReader(bam)
                res = encode_location(bam, fasta, loc, opt, reader)
                rec = finish_record(res, loc, opt, dtype) if res is not None else None
            if rec is None:
                errors += 1
                continue
            out[n] = rec
            n += 1
    finally:
        if bam is not None:
            bam.close()
            fasta.close()
    return out[:n], errors
