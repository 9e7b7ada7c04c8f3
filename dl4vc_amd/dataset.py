"""Candidate-site assembly: HDF5 record -> the six uint8 planes the device consumes (row A2).

Restates the inference slice of ``ContextDatasetFromNumpy._get_generator`` (reference:
dl4vc/dataset.py:500-680) and ``sample_single_reads`` (:256-287) for the supported flags
(no augmentation, no dynamic down-sampling):

* rows ``[start, start+200)`` of ``single_reads`` / ``q-scores`` / ``strand`` with
  ``start = max(0, int(max(num_reads, 200)/2 - 100))`` (dataset.py:517-521) -- 0 for every file the
  converter can write;
* if ``num_reads <= R`` the first R rows are used, otherwise a random SORTED subset of R of the first
  ``num_reads`` rows, the same subset for all three planes (dataset.py:274-281).  The reference draws it
  from the unseeded global numpy RNG, so its inference is not reproducible for deep pileups; here the
  generator is an explicit, seedable input (``rng``) and the draw sequence is the reference's (one
  ``random()`` for the disabled dynamic-down-sampling coin, dataset.py:531, then one ``choice``), so that
  ``np.random.RandomState(s)`` here reproduces ``np.random.seed(s)`` there;
* ``ref = ref_bases``; allele masks from ``dl4vc_amd.alleles`` with the blacklist fallback.

Output keeps the HDF5-native ``[read][pos]`` order -- the reference's ``(pos, read)`` transpose
(dataset.py:521) is an artefact of its NCHW plumbing that the device path does not need.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from .alleles import safe_allele_mask_vectors, parse_candidate
from .hdf5_schema import STORE_MAX_READS
from .synth import SiteBatch


@dataclass
class Site:
    reads: np.ndarray       # (R,L) u8
    qual: np.ndarray
    strand: np.ndarray
    ref: np.ndarray         # (L,) u8
    ref_mask: np.ndarray
    var_mask: np.ndarray
    vcfrec: str
    name: str
    num_reads: int
    rows: np.ndarray        # which stored rows were used (the recorded permutation)
    blacklist: bool


def _scalar(x) -> int:
    return int(np.asarray(x).reshape(-1)[0])


def select_rows(num_reads: int, stored_rows: int, max_reads: int, rng=None) -> np.ndarray:
    """Row subset of ``sample_single_reads`` (dataset.py:256-287) with random=True and no down-sampling."""
    max_reads = min(max_reads, stored_rows)
    if max_reads >= num_reads:
        return np.arange(max_reads)
    pool = min(stored_rows, num_reads)
    if rng is None:
        raise ValueError("site stores %d reads > %d: an explicit rng is required to pin the read subset"
                         % (num_reads, max_reads))
    return np.sort(rng.choice(pool, min(max_reads, num_reads), replace=False))


def assemble_site(record, max_reads: int, rng=None, use_q: bool = True, use_strand: bool = True,
                  store_max_reads: int = STORE_MAX_READS) -> Site:
    """One structured record (``hdf5_schema.record_dtype``) -> ``Site``."""
    num_reads = _scalar(record["num_reads"])
    mid = int(max(num_reads, store_max_reads) / 2)
    start = max(0, int(mid - store_max_reads / 2))
    rows_all = np.asarray(record["single_reads"], dtype=np.uint8)[start:start + store_max_reads]
    if rng is not None:
        rng.random_sample()          # the reference's disabled dynamic-down-sampling coin (dataset.py:531)
    rows = select_rows(num_reads, rows_all.shape[0], max_reads, rng)
    if len(rows) < max_reads and num_reads > len(rows):
        # num_reads > 2 x stored rows: the window [start, start + store) runs past the stored rows and fewer than
        # max_reads remain.  The reference yields a (L, k) item its DataLoader cannot collate (dataset.py:517-521,
        # :270-281); the native loader (csrc/dan_loader.cpp::build_batch) refuses the same way.
        raise ValueError("num_reads %d leaves %d stored rows in the sampling window (< %d reads): the reference cannot "
                         "batch this site either" % (num_reads, len(rows), max_reads))
    reads = np.ascontiguousarray(rows_all[rows])
    if use_q:
        qual = np.ascontiguousarray(np.asarray(record["q-scores"], np.uint8)[start:start + store_max_reads][rows])
    else:
        qual = np.zeros_like(reads)
    if use_strand:
        strand = np.ascontiguousarray(np.asarray(record["strand"], np.uint8)[start:start + store_max_reads][rows])
    else:
        strand = np.zeros_like(reads)
    ref = np.asarray(record["ref_bases"], dtype=np.uint8).copy()
    vcfrec = bytes(record["vcfrec"]).rstrip(b"\x00").decode()
    name = bytes(record["name"]).rstrip(b"\x00").decode()
    parse_candidate(vcfrec)          # the reference parses (and may raise) here, dataset.py:584-585
    ref_mask, var_mask, black = safe_allele_mask_vectors(vcfrec, ref)
    return Site(reads, qual, strand, ref, ref_mask, var_mask, vcfrec, name, num_reads, rows, black)


def assemble_batch(records, max_reads: int, seed: Optional[int] = None, **kw) -> SiteBatch:
    """Records -> contiguous ``SiteBatch``.  ``seed`` pins the read subsets of deep pileups; site i uses
    ``RandomState(seed + i)`` so a shard boundary does not change any site's subset."""
    sites: List[Site] = []
    for i, rec in enumerate(records):
        rng = np.random.RandomState(seed + i) if seed is not None else None
        sites.append(assemble_site(rec, max_reads, rng, **kw))
    stack = lambda f: np.stack([getattr(s, f) for s in sites])   # noqa: E731
    return SiteBatch(stack("reads"), stack("qual"), stack("strand"), stack("ref"), stack("ref_mask"),
                     stack("var_mask"), [s.vcfrec for s in sites],
                     np.array([s.num_reads for s in sites], np.int32))
