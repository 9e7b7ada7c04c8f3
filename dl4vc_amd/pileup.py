"""Pileup columns over a window of a coordinate-sorted BAM, and their samtools-style read strings (row N4).

Stands where ``samfile.pileup(truncate=True, contig=..., start=..., stop=..., ignore_overlaps=False, stepper="nofilter",
min_base_quality=q, min_mapping_quality=0)`` and ``PileupColumn.get_query_sequences(mark_ends=True, add_indels=True,
mark_matches=True)`` / ``get_query_qualities()`` stand in the reference (``tools/convert_bam_single_reads.py:873-874,922,987``).
pysam / htslib are absent from the image; what is restated here is htslib's published pileup algorithm (``bam_plp`` and the
per-read CIGAR resolution behind ``bam_pileup1_t``):

* reads flagged unmapped / secondary / QC-fail / duplicate never enter the pileup (``BAM_DEF_MASK``); reads without a
  reference-consuming operation are skipped;
* a column is produced for every reference position of ``[start, stop)`` covered by at least one read (a deletion or a
  reference skip covers its positions); its entries are the covering reads in file order;
* per entry: ``qpos`` (query index of the base at the column; for a deletion / skip the index of the next query base),
  ``is_del``, ``is_refskip``, ``indel`` (> 0: length of the insertion that follows this column, consecutive ``I`` merged
  across ``P``; < 0: minus the length of the deletion that starts after it, consecutive ``D`` merged; only reported on the
  last column of the current operation, and not for a deletion continuing a deletion), ``is_head`` / ``is_tail`` (first /
  last reference position of the read);
* the read string: ``^`` + chr(min(mapq, 93) + 33) on the head column, the base (upper case forward, lower case reverse;
  with the "nofilter" stepper pysam holds no reference sequence, so ``mark_matches`` never produces ``.`` / ``,`` -- the
  reference's own tables have no entry for them, tools/convert_bam_single_reads.py:50-56), ``*`` for a deletion, ``>`` / ``<``
  for a reference skip, ``+<n><bases>`` / ``-<n><N...>``, ``$`` on the tail column;
* ``max_depth`` (pysam's default 8000): a read starting at the position the iterator stands on is dropped once more than
  that many reads are active.

PARITY UNPINNED (no htslib here): tests hold this module to hand-derived columns of the SAM specification's example
alignment and to the rules above.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, Iterator, List, Optional

import numpy as np

from .bamio import (BamRecord, CDEL, CDIFF, CEQUAL, CINS, CMATCH, CPAD, CREF_SKIP, CSOFT_CLIP, FDUP, FQCFAIL, FSECONDARY,
                    FUNMAP)

DEFAULT_FLAG_MASK = FUNMAP | FSECONDARY | FQCFAIL | FDUP
_ALIGNED = (CMATCH, CEQUAL, CDIFF)
_REF_OPS = (CMATCH, CEQUAL, CDIFF, CDEL, CREF_SKIP)


class ReadTrack:
    """One read resolved against the reference: per covered position ``qpos`` / ``is_del`` / ``is_refskip`` / ``indel``."""

    __slots__ = ("rec", "start", "end", "qpos", "is_del", "is_refskip", "indel", "key")

    def __init__(self, rec: BamRecord):
        self.rec = rec
        self.start = rec.pos
        cig = rec.cigar
        n = sum(l for op, l in cig if op in _REF_OPS)
        self.end = rec.pos + n
        qpos = np.zeros(n, np.int32)
        is_del = np.zeros(n, bool)
        is_skip = np.zeros(n, bool)
        indel = np.zeros(n, np.int32)
        x = y = 0
        for k, (op, l) in enumerate(cig):
            if op in _ALIGNED:
                qpos[x:x + l] = np.arange(y, y + l)
                x += l
                y += l
            elif op in (CDEL, CREF_SKIP):
                qpos[x:x + l] = y
                is_del[x:x + l] = True
                is_skip[x:x + l] = op == CREF_SKIP
                x += l
            elif op in (CINS, CSOFT_CLIP):
                y += l
            if op in _REF_OPS and x > 0 and k + 1 < len(cig):       # what follows the LAST column of this operation
                op2, l2 = cig[k + 1]
                v = 0
                if op2 == CDEL and op != CDEL:
                    v = -l2
                    for op3, l3 in cig[k + 2:]:
                        if op3 != CDEL:
                            break
                        v -= l3
                elif op2 == CINS:
                    v = l2
                    for op3, l3 in cig[k + 2:]:
                        if op3 == CINS:
                            v += l3
                        elif op3 != CPAD:
                            break
                elif op2 == CPAD and k + 2 < len(cig):
                    for op3, l3 in cig[k + 2:]:
                        if op3 == CINS:
                            v += l3
                        elif op3 in _REF_OPS:
                            break
                indel[x - 1] = v
        self.qpos, self.is_del, self.is_refskip, self.indel = qpos, is_del, is_skip, indel
        self.key = "%s:%s" % (rec.name, rec.seq)                    # tools/convert_bam_single_reads.py:991


@dataclass
class PileupEntry:
    track: ReadTrack
    qpos: int
    is_del: bool
    is_refskip: bool
    indel: int
    is_head: bool
    is_tail: bool


@dataclass
class PileupColumn:
    reference_pos: int
    entries: List[PileupEntry]

    def query_sequences(self, min_base_quality: int = 0, reference: Optional[str] = None, reference_start: int = 0) -> List[str]:
        """``get_query_sequences(mark_ends=True, add_indels=True, mark_matches=True)`` without a reference sequence."""
        out = []
        for e in self.entries:
            rec = e.track.rec
            if _quality(e) < min_base_quality:
                continue
            rev = rec.is_reverse
            buf = []
            if e.is_head:
                buf.append("^" + ("~" if rec.mapq > 93 else chr(rec.mapq + 33)))
            if not e.is_del:
                c = rec.seq[e.qpos] if e.qpos < len(rec.seq) else "N"
                buf.append(c.lower() if rev else c.upper())
            elif e.is_refskip:
                buf.append("<" if rev else ">")
            else:
                buf.append("*")
            if e.indel > 0:
                ins = "".join(rec.seq[e.qpos + j] if e.qpos + j < len(rec.seq) else "N" for j in range(1, e.indel + 1))
                buf.append("+%d%s" % (e.indel, ins.lower() if rev else ins.upper()))
            elif e.indel < 0:
                n = -e.indel
                if reference is not None:
                    o = self.reference_pos + 1 - reference_start
                    gone = reference[o:o + n].ljust(n, "N")
                else:
                    gone = "N" * n
                buf.append("-%d%s" % (n, gone.lower() if rev else gone.upper()))
            if e.is_tail:
                buf.append("$")
            out.append("".join(buf))
        return out

    def query_qualities(self, min_base_quality: int = 0) -> List[int]:
        return [q for q in (_quality(e) for e in self.entries) if q >= min_base_quality]

    def query_ids(self, min_base_quality: int = 0) -> List[str]:
        return [e.track.key for e in self.entries if _quality(e) >= min_base_quality]


def _quality(e: PileupEntry) -> int:
    q = e.track.rec.qual
    return int(q[e.qpos]) if e.qpos < len(q) else 0


def resolve_reads(reads: Iterable[BamRecord], flag_mask: int = DEFAULT_FLAG_MASK) -> List[ReadTrack]:
    """The reads that enter a pileup, resolved against the reference, in file order."""
    tracks: List[ReadTrack] = []
    for rec in reads:
        if rec.flag & flag_mask or rec.tid < 0:
            continue
        if not any(op in _REF_OPS for op, _ in rec.cigar):
            continue
        tracks.append(ReadTrack(rec))
    return tracks


def pileup_columns(reads: Iterable[BamRecord], start: int, stop: int, flag_mask: int = DEFAULT_FLAG_MASK,
                   max_depth: int = 8000, tracks: Optional[List[ReadTrack]] = None) -> Iterator[PileupColumn]:
    """Columns of ``[start, stop)`` (0-based) from ``reads``: the records of one reference overlapping the window, in file
    (coordinate) order (or from ``tracks`` = ``resolve_reads(reads)``)."""
    if tracks is None:
        tracks = resolve_reads(reads, flag_mask)
    active: List[ReadTrack] = []
    nxt = 0
    pos = min((t.start for t in tracks), default=stop)
    while pos < stop:
        while nxt < len(tracks) and tracks[nxt].start <= pos:
            t = tracks[nxt]
            nxt += 1
            if len(active) > max_depth and t.start == pos:
                continue
            active.append(t)
        if active:
            active = [t for t in active if t.end > pos]
        if not active:
            if nxt >= len(tracks):
                return
            pos = tracks[nxt].start
            continue
        if pos >= start:
            entries = []
            for t in active:
                i = pos - t.start
                entries.append(PileupEntry(t, int(t.qpos[i]), bool(t.is_del[i]), bool(t.is_refskip[i]), int(t.indel[i]),
                                           pos == t.start, pos == t.end - 1))
            yield PileupColumn(pos, entries)
        pos += 1
