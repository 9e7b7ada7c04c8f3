"""BAM / BGZF / BAI and FASTA readers for the pileup encoder.

The reference reads alignments through pysam, i.e. htslib (``tools/convert_bam_single_reads.py:871-874``:
``pysam.AlignmentFile(samfile, "rb")``, ``pysam.FastaFile(fasta)``); neither is in this image, so the formats are read here
from their published specification (SAM/BAM format specification v1, sections 4.1 BGZF, 4.2 BAM, 5.2 BAI; faidx ``.fai``):

* BGZF: gzip members of at most 64 KiB with a ``BC`` extra field holding the block size; a virtual file offset is
  ``block_start << 16 | offset_in_block``;
* BAM: header text + reference dictionary, then length-prefixed alignment records (4-bit packed sequence, CIGAR as
  ``len << 4 | op`` with ops ``MIDNSHP=X``);
* BAI: per reference a binning index and a 16-kbp linear index of the smallest virtual offset of any alignment overlapping
  the window -- ``fetch`` seeks to the linear-index offset of the window holding ``start`` and scans forward;
* FASTA + ``.fai`` (name, length, offset, bases per line, bytes per line); the index is built in memory when the file has none.

PARITY UNPINNED for this module: there is no htslib, samtools or pysam in the image to read the same files with.  The tests
hold the readers to the specification's own layout rules through an independent writer (``BamWriter``), to hand-packed
records, and to each other (indexed fetch = linear scan).
"""
from __future__ import annotations

import io
import os
import struct
import zlib
from dataclasses import dataclass
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from .vcfpost import BgzfWriter, reg2bin

BAM_MAGIC = b"BAM\x01"
BAI_MAGIC = b"BAI\x01"
CIGAR_OPS = "MIDNSHP=X"
SEQ_CODES = "=ACMGRSVTWYHKDBN"
# flag bits (SAM specification section 1.4)
FPAIRED, FPROPER_PAIR, FUNMAP, FMUNMAP, FREVERSE, FMREVERSE, FREAD1, FREAD2, FSECONDARY, FQCFAIL, FDUP, FSUPPLEMENTARY = (
    1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048)
# CIGAR op codes
CMATCH, CINS, CDEL, CREF_SKIP, CSOFT_CLIP, CHARD_CLIP, CPAD, CEQUAL, CDIFF = range(9)
_CONSUMES_QUERY = (True, True, False, False, True, False, False, True, True)
_CONSUMES_REF = (True, False, True, True, False, False, False, True, True)

_SEQ_LUT = np.array([ord(c) for c in SEQ_CODES], np.uint8)


class BgzfReader:
    """Sequential BGZF reader with ``seek`` / ``tell`` on virtual offsets."""

    def __init__(self, path: str):
        self.f = open(path, "rb")
        self.block_start = 0          # file offset of the block in ``self.data``
        self.next_block = 0           # file offset of the block after it
        self.data = b""
        self.off = 0

    def close(self):
        self.f.close()

    def _load(self, file_off: int) -> bool:
        self.f.seek(file_off)
        head = self.f.read(18)
        if len(head) == 0:
            self.block_start, self.next_block, self.data, self.off = file_off, file_off, b"", 0
            return False
        if len(head) < 18 or head[:4] != b"\x1f\x8b\x08\x04":
            raise ValueError("not a BGZF block at file offset %d" % file_off)
        xlen = struct.unpack_from("<H", head, 10)[0]
        extra = head[12:] + self.f.read(xlen - 6)
        bsize = None
        i = 0
        while i + 4 <= len(extra):
            slen = struct.unpack_from("<H", extra, i + 2)[0]
            if extra[i:i + 2] == b"BC":
                bsize = struct.unpack_from("<H", extra, i + 4)[0]
            i += 4 + slen
        if bsize is None:
            raise ValueError("BGZF block without a BC field at file offset %d" % file_off)
        body = self.f.read(bsize + 1 - 12 - xlen)
        if len(body) < 8:
            raise ValueError("truncated BGZF block at file offset %d" % file_off)
        data = zlib.decompress(body[:-8], -15)
        crc, isize = struct.unpack("<II", body[-8:])
        if len(data) != isize or (zlib.crc32(data) & 0xffffffff) != crc:
            raise ValueError("BGZF block at file offset %d fails its CRC / size check" % file_off)
        self.block_start, self.next_block, self.data, self.off = file_off, file_off + bsize + 1, data, 0
        return True

    def tell(self) -> int:
        return (self.block_start << 16) | self.off

    def seek(self, voffset: int) -> None:
        blk, off = voffset >> 16, voffset & 0xffff
        if blk != self.block_start or not self.data:
            self._load(blk)
        self.off = off

    def read(self, n: int) -> bytes:
        out = []
        while n > 0:
            if self.off >= len(self.data):
                if not self._load(self.next_block):
                    break
                continue
            take = self.data[self.off:self.off + n]
            out.append(take)
            self.off += len(take)
            n -= len(take)
        return b"".join(out)


@dataclass
class BamRecord:
    tid: int
    pos: int                      # 0-based leftmost reference position
    mapq: int
    flag: int
    name: str
    cigar: Tuple[Tuple[int, int], ...]      # (op, length)
    seq: str                      # ASCII bases as stored (forward reference strand)
    qual: np.ndarray              # uint8 phred, 255 = absent
    next_tid: int = -1
    next_pos: int = -1
    tlen: int = 0
    aux: bytes = b""

    @property
    def is_reverse(self) -> bool:
        return bool(self.flag & FREVERSE)

    @property
    def reference_end(self) -> int:
        """One past the last reference position the alignment covers (``bam_endpos``: a read without reference-consuming
        operations covers one position)."""
        n = sum(l for op, l in self.cigar if _CONSUMES_REF[op])
        return self.pos + (n if n > 0 else 1)

    def cigar_string(self) -> str:
        return "".join("%d%s" % (l, CIGAR_OPS[op]) for op, l in self.cigar) or "*"


def parse_record(buf: bytes) -> BamRecord:
    """One alignment record (the bytes behind its ``block_size`` field; specification section 4.2)."""
    tid, pos, l_name, mapq, _bin, n_cig, flag, l_seq, ntid, npos, tlen = struct.unpack_from("<iiBBHHHiiii", buf, 0)
    o = 32
    name = buf[o:o + l_name - 1].decode("ascii", "replace")
    o += l_name
    raw = np.frombuffer(buf, "<u4", n_cig, o)
    cigar = tuple((int(v & 0xf), int(v >> 4)) for v in raw)
    o += 4 * n_cig
    packed = np.frombuffer(buf, np.uint8, (l_seq + 1) // 2, o)
    codes = np.empty(2 * len(packed), np.uint8)
    codes[0::2] = packed >> 4
    codes[1::2] = packed & 0xf
    seq = _SEQ_LUT[codes[:l_seq]].tobytes().decode("ascii")
    o += (l_seq + 1) // 2
    qual = np.frombuffer(buf, np.uint8, l_seq, o).copy()
    o += l_seq
    return BamRecord(tid, pos, mapq, flag, name, cigar, seq, qual, ntid, npos, tlen, bytes(buf[o:]))


class BamFile:
    """Header + sequential / indexed access to the alignments of one BAM file."""

    def __init__(self, path: str, index: Optional[str] = None):
        self.path = path
        self.r = BgzfReader(path)
        if self.r.read(4) != BAM_MAGIC:
            raise ValueError("%s is not a BAM file" % path)
        l_text = struct.unpack("<i", self.r.read(4))[0]
        self.header_text = self.r.read(l_text).split(b"\x00", 1)[0].decode("utf-8", "replace")
        n_ref = struct.unpack("<i", self.r.read(4))[0]
        self.references: List[str] = []
        self.lengths: List[int] = []
        for _ in range(n_ref):
            l_name = struct.unpack("<i", self.r.read(4))[0]
            self.references.append(self.r.read(l_name)[:-1].decode("ascii"))
            self.lengths.append(struct.unpack("<i", self.r.read(4))[0])
        self.first_record = self.r.tell()
        self._tid = {n: i for i, n in enumerate(self.references)}
        self.index = None
        for cand in ([index] if index else [path + ".bai", os.path.splitext(path)[0] + ".bai"]):
            if cand and os.path.isfile(cand):
                self.index = BaiIndex.load(cand)
                break

    def close(self):
        self.r.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def get_tid(self, name: str) -> int:
        """Reference id of a contig name, trying the other ``chr`` spelling as well; -1 when absent."""
        for n in (name, name[3:] if name.startswith("chr") else "chr" + name):
            if n in self._tid:
                return self._tid[n]
        return -1

    def _next(self) -> Optional[Tuple[int, BamRecord]]:
        at = self.r.tell()
        head = self.r.read(4)
        if len(head) < 4:
            return None
        size = struct.unpack("<i", head)[0]
        body = self.r.read(size)
        if len(body) < size:
            raise ValueError("truncated BAM record in %s" % self.path)
        return at, parse_record(body)

    def __iter__(self) -> Iterator[BamRecord]:
        self.r.seek(self.first_record)
        while True:
            nx = self._next()
            if nx is None:
                return
            yield nx[1]

    def fetch(self, tid: int, start: int, stop: int) -> Iterator[BamRecord]:
        """Alignments on reference ``tid`` overlapping ``[start, stop)`` of a coordinate-sorted file, in file order
        (unmapped records placed on the reference are included, as ``bam_itr`` includes them)."""
        if tid < 0:
            return
        at = self.first_record
        if self.index is not None:
            off = self.index.linear_offset(tid, start)
            if off is None:
                return
            at = off
        self.r.seek(at)
        while True:
            nx = self._next()
            if nx is None:
                return
            rec = nx[1]
            if rec.tid != tid:
                if rec.tid > tid or rec.tid < 0:
                    return
                continue
            if rec.pos >= stop:
                return
            if rec.reference_end > start:
                yield rec


class WindowReader:
    """``fetch`` for a run of queries that move forward along a reference (candidate locations in VCF order): the records
    of the previous window are kept and the file is read on from where the last query stopped, so every alignment is
    inflated and parsed once instead of once per overlapping window (an indexed seek lands on a 16-kbp boundary: at 30x
    coverage that is up to 3 000 records of lead-in per query).  A query that moves backwards, changes reference or jumps
    more than ``max_gap`` ahead starts over through the index."""

    def __init__(self, bam: BamFile, max_gap: int = 1 << 16):
        self.bam, self.max_gap = bam, max_gap
        self.tid, self.last_start, self.scanned_to = -2, -1, -1
        self.kept: List[BamRecord] = []
        self.pending: Optional[BamRecord] = None
        self.at = 0                   # virtual offset behind the last record taken from the file
        self.eof = True

    def reads(self, tid: int, start: int, stop: int) -> List[BamRecord]:
        bam = self.bam
        if tid < 0:
            return []
        if tid != self.tid or start < self.last_start or start > self.scanned_to + self.max_gap:
            self.kept, self.pending, self.tid, self.scanned_to, self.eof = [], None, tid, -1, False
            self.at = bam.first_record
            if bam.index is not None:
                off = bam.index.linear_offset(tid, start)
                if off is None:
                    self.eof = True
                else:
                    self.at = off
        else:
            self.kept = [r for r in self.kept if r.reference_end > start]
        self.last_start = start
        if stop > self.scanned_to and not self.eof:
            rec, self.pending = self.pending, None
            if rec is None:
                bam.r.seek(self.at)
            while True:
                if rec is None:
                    got = bam._next()
                    if got is None:
                        self.eof = True
                        break
                    rec = got[1]
                if rec.tid != tid:
                    if rec.tid > tid or rec.tid < 0:
                        self.eof = True
                        break
                elif rec.pos >= stop:
                    self.pending = rec
                    break
                elif rec.reference_end > start:
                    self.kept.append(rec)
                rec = None
            self.at = bam.r.tell()
            self.scanned_to = stop
        return [r for r in self.kept if r.pos < stop and r.reference_end > start]


# ------------------------------------------------------------------------------------------------------
# BAI (specification section 5.2)
# ------------------------------------------------------------------------------------------------------
class BaiIndex:
    def __init__(self, bins: List[Dict[int, List[Tuple[int, int]]]], linear: List[List[int]]):
        self.bins, self.linear = bins, linear

    @classmethod
    def load(cls, path: str) -> "BaiIndex":
        raw = open(path, "rb").read()
        if raw[:4] != BAI_MAGIC:
            raise ValueError("%s is not a BAI index" % path)
        o = 4
        n_ref = struct.unpack_from("<i", raw, o)[0]
        o += 4
        bins, linear = [], []
        for _ in range(n_ref):
            n_bin = struct.unpack_from("<i", raw, o)[0]
            o += 4
            b: Dict[int, List[Tuple[int, int]]] = {}
            for _ in range(n_bin):
                bid, n_chunk = struct.unpack_from("<Ii", raw, o)
                o += 8
                b[bid] = [struct.unpack_from("<QQ", raw, o + 16 * k) for k in range(n_chunk)]
                o += 16 * n_chunk
            n_intv = struct.unpack_from("<i", raw, o)[0]
            o += 4
            linear.append(list(struct.unpack_from("<%dQ" % n_intv, raw, o)))
            o += 8 * n_intv
            bins.append(b)
        return cls(bins, linear)

    def linear_offset(self, tid: int, start: int) -> Optional[int]:
        """Virtual offset from which a forward scan sees every alignment overlapping ``start`` or later; None when the
        reference holds no alignment at or after the window."""
        if tid >= len(self.linear):
            return None
        lin = self.linear[tid]
        w = max(start, 0) >> 14
        if w < len(lin):
            for v in lin[w:]:
                if v:
                    return v
            return None
        return None

    def save(self, path: str) -> None:
        out = io.BytesIO()
        out.write(BAI_MAGIC + struct.pack("<i", len(self.bins)))
        for b, lin in zip(self.bins, self.linear):
            out.write(struct.pack("<i", len(b)))
            for bid in sorted(b):
                out.write(struct.pack("<Ii", bid, len(b[bid])))
                for beg, end in b[bid]:
                    out.write(struct.pack("<QQ", beg, end))
            out.write(struct.pack("<i", len(lin)))
            out.write(struct.pack("<%dQ" % len(lin), *lin))
        open(path, "wb").write(out.getvalue())


def build_bai(bam_path: str, out_path: Optional[str] = None) -> BaiIndex:
    """Index a coordinate-sorted BAM (what ``samtools index`` writes): bins with merged adjacent chunks, linear index with
    the gaps filled from the left."""
    with BamFile(bam_path, index="") as bam:
        bam.index = None
        n = len(bam.references)
        bins: List[Dict[int, List[Tuple[int, int]]]] = [dict() for _ in range(n)]
        linear: List[List[int]] = [[] for _ in range(n)]
        bam.r.seek(bam.first_record)
        last = (-1, -1)
        while True:
            nx = bam._next()
            if nx is None:
                break
            at, rec = nx
            end_v = bam.r.tell()
            if rec.tid < 0:
                continue
            if (rec.tid, rec.pos) < last:
                raise ValueError("%s is not coordinate-sorted: cannot index" % bam_path)
            last = (rec.tid, rec.pos)
            beg, end = rec.pos, rec.reference_end
            chunks = bins[rec.tid].setdefault(reg2bin(beg, end), [])
            if chunks and chunks[-1][1] == at:
                chunks[-1] = (chunks[-1][0], end_v)
            else:
                chunks.append((at, end_v))
            lin = linear[rec.tid]
            w0, w1 = beg >> 14, (end - 1) >> 14
            if len(lin) <= w1:
                lin.extend([0] * (w1 + 1 - len(lin)))
            for w in range(w0, w1 + 1):
                if lin[w] == 0:
                    lin[w] = at
        for lin in linear:                                   # empty windows take the next alignment's offset... from the left
            prev = 0
            for i, v in enumerate(lin):
                if v == 0:
                    lin[i] = prev
                else:
                    prev = v
    idx = BaiIndex(bins, linear)
    if out_path is not None:
        idx.save(out_path)
    return idx


# ------------------------------------------------------------------------------------------------------
# BAM writer (tests, tools/make_test_bam.py): the inverse of parse_record, kept independent of it
# ------------------------------------------------------------------------------------------------------
def pack_record(tid: int, pos: int, name: str, flag: int, mapq: int, cigar: Sequence[Tuple[int, int]], seq: str,
                qual: Optional[Sequence[int]] = None, next_tid: int = -1, next_pos: int = -1, tlen: int = 0) -> bytes:
    l_seq = len(seq)
    end = pos + (sum(l for op, l in cigar if _CONSUMES_REF[op]) or 1)
    nm = name.encode("ascii") + b"\x00"
    out = bytearray(struct.pack("<iiBBHHHiiii", tid, pos, len(nm), mapq, reg2bin(pos, end), len(cigar), flag, l_seq,
                                next_tid, next_pos, tlen))
    out += nm
    for op, l in cigar:
        out += struct.pack("<I", (l << 4) | op)
    codes = [SEQ_CODES.index(c.upper()) if c.upper() in SEQ_CODES else 15 for c in seq]
    if l_seq & 1:
        codes.append(0)
    out += bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))
    out += bytes(qual) if qual is not None else b"\xff" * l_seq
    return struct.pack("<i", len(out)) + bytes(out)


class BamWriter:
    def __init__(self, path: str, references: Sequence[Tuple[str, int]], header_text: Optional[str] = None, level: int = 6):
        self.f = open(path, "wb")
        self.w = BgzfWriter(self.f, level)
        text = header_text if header_text is not None else "@HD\tVN:1.6\tSO:coordinate\n" + "".join(
            "@SQ\tSN:%s\tLN:%d\n" % r for r in references)
        t = text.encode()
        self.w.write(BAM_MAGIC + struct.pack("<i", len(t)) + t + struct.pack("<i", len(references)))
        for name, length in references:
            nm = name.encode() + b"\x00"
            self.w.write(struct.pack("<i", len(nm)) + nm + struct.pack("<i", length))

    def write(self, *args, **kw) -> None:
        self.w.write(pack_record(*args, **kw))

    def close(self) -> None:
        self.w.close()
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


# ------------------------------------------------------------------------------------------------------
# FASTA + .fai
# ------------------------------------------------------------------------------------------------------
class FastaFile:
    """``pysam.FastaFile``'s ``references`` / ``fetch(reference, start, end)`` (0-based half-open; bases as stored)."""

    def __init__(self, path: str):
        self.path = path
        self.f = open(path, "rb")
        self.index: Dict[str, Tuple[int, int, int, int]] = {}
        fai = path + ".fai"
        if os.path.isfile(fai):
            for line in open(fai):
                p = line.rstrip("\n").split("\t")
                if len(p) >= 5:
                    self.index[p[0]] = (int(p[1]), int(p[2]), int(p[3]), int(p[4]))
        else:
            self._scan()
        self.references = list(self.index)

    def _scan(self) -> None:
        name, length, offset, lb, lw = None, 0, 0, 0, 0
        pos = 0
        for line in self.f:
            if line.startswith(b">"):
                if name is not None:
                    self.index[name] = (length, offset, lb, lw)
                name = line[1:].split()[0].decode()
                length, offset, lb, lw = 0, pos + len(line), 0, 0
            elif name is not None:
                bases = len(line.rstrip(b"\r\n"))
                if lb == 0:
                    lb, lw = bases, len(line)
                length += bases
            pos += len(line)
        if name is not None:
            self.index[name] = (length, offset, lb, lw)

    def close(self):
        self.f.close()

    def _entry(self, reference: str):
        for n in (reference, reference[3:] if reference.startswith("chr") else "chr" + reference):
            if n in self.index:
                return self.index[n]
        raise KeyError("sequence '%s' not present in %s" % (reference, self.path))

    def get_reference_length(self, reference: str) -> int:
        return self._entry(reference)[0]

    def fetch(self, reference: str, start: int, end: int) -> str:
        length, offset, lb, lw = self._entry(reference)
        start, end = max(0, start), min(end, length)
        if end <= start:
            return ""
        first = offset + (start // lb) * lw + start % lb
        last = offset + ((end - 1) // lb) * lw + (end - 1) % lb
        self.f.seek(first)
        raw = self.f.read(last - first + 1)
        return raw.replace(b"\n", b"").replace(b"\r", b"").decode("ascii")
