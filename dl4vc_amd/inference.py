"""Inference harness: HDF5 candidates -> scored VCF records.

Stands where ``trainer.test`` stands in the reference (dl4vc/trainer.py:474-681) for the inference-only
run of ``main.py`` (main.py:213-218): iterate the candidate records in order, assemble batches, call the
model, write ``BP/NV/HV/OV`` into the VCF ID column.  Everything the reference's loop does that does not
reach the VCF (loss logging, ROC/PR prints on the all-'FP' labels, trust-region weighting) is omitted.
"""
from __future__ import annotations

import sys
import time
from typing import Callable, Optional

import numpy as np

from .dataset import assemble_batch
from .hdf5io import CandidateFile
from .shard import shard_range
from .vcf import scored_record, threshold_distance, FormatOptions, PIPELINE_OPTIONS


NEAR_EPS = 1e-4


class _Pipeline:
    """One batch of lookahead over ``net``: batch k+1 is enqueued (``forward_u8_async``: pinned staging, H2D on a copy
    stream) before batch k's scores are awaited and formatted, so the VCF text formatting -- the reference formats each
    score with ``'%.8f' % tensor``, one D2H sync per scalar, utils.py:168-178 -- and the loader hand-off overlap the
    forward.  A model without the asynchronous pair (the CPU test double) is called synchronously."""

    def __init__(self, net, write, use_var_type_threshold, stats=None):
        self.net, self.write, self.vt_thr = net, write, use_var_type_threshold
        self.stats = stats
        self.pending = None
        self.async_ok = hasattr(net, "forward_u8_async")
        self.max_batch = net.handle.query("max_batch") if self.async_ok else 0

    def _emit(self, batch, out):
        vt = out["vt_prob"]
        bp = (1.0 - vt[:, 0]) if self.vt_thr else out["bp"]                   # trainer.py:611-621
        self.write("".join(scored_record(r, b, v) + "\n" for r, b, v in zip(batch.vcfrec, bp, vt)))
        if self.stats is not None:
            # sites whose scores sit within 1e-4 (the score tolerance of the parity gate) of a genotype threshold of the published
            # pipeline (call_variants.sh:154-160): where a call could differ between two correct evaluations
            d = threshold_distance(batch.vcfrec, vt, FormatOptions(**PIPELINE_OPTIONS))
            self.stats["near_threshold"] = self.stats.get("near_threshold", 0) + int((d < NEAR_EPS).sum())
            self.stats["sites"] = self.stats.get("sites", 0) + len(batch.vcfrec)

    def submit(self, batch):
        if self.async_ok and len(batch.vcfrec) <= self.max_batch:
            token = self.net.forward_u8_async(*batch.arrays())
            self.drain()
            self.pending = (batch, token)
        else:
            self.drain()
            self._emit(batch, self.net.forward_u8(*batch.arrays()))

    def drain(self):
        if self.pending is not None:
            batch, token = self.pending
            self.pending = None
            self._emit(batch, self.net.wait(token))


def score_records(net, source: CandidateFile, write: Callable[[str], None], lo: int = 0, hi: Optional[int] = None,
                  sites_per_launch: int = 4096, reads_seed: int = 0, use_var_type_threshold: bool = False,
                  log=None, stats=None) -> int:
    """Score records ``[lo, hi)`` of ``source`` with ``net`` (anything with ``forward_u8`` and ``config``) and hand
    each scored VCF line (with '\\n') to ``write``.  Returns the number of sites scored."""
    cfg = net.config
    hi = len(source) if hi is None else min(hi, len(source))
    done = 0
    t0 = time.perf_counter()
    pipe = _Pipeline(net, write, use_var_type_threshold, stats)
    for b0 in range(lo, hi, sites_per_launch):
        recs = source.read(b0, min(b0 + sites_per_launch, hi))
        # the seed is tied to the ABSOLUTE record index, so shard boundaries never change a site's read subset
        batch = assemble_batch(recs, cfg.reads, seed=reads_seed + b0, use_q=cfg.use_q, use_strand=cfg.use_strand)
        pipe.submit(batch)
        done += len(recs)
        if log:
            dt = time.perf_counter() - t0
            log("  submitted %d/%d sites (%.0f sites/s)" % (done, hi - lo, done / max(dt, 1e-9)))
    pipe.drain()
    return done


def score_file_native(net, hdf_path: str, write: Callable[[str], None], lo: int, hi: int, sites_per_launch: int = 4096,
                      reads_seed: int = 0, use_var_type_threshold: bool = False, log=None,
                      threads: int = 0, stats=None) -> int:
    """Same contract as ``score_records`` but fed by the native batched loader (dl4vc_amd/loader.py): chunk
    inflate and site assembly of batch k+1.. run in C++ threads while the GPU scores batch k."""
    import os
    from .loader import NativeLoader
    cfg = net.config
    threads = threads or max(2, min(16, (os.cpu_count() or 4)))
    done = 0
    t0 = time.perf_counter()
    with NativeLoader(hdf_path, cfg.reads, batch_sites=sites_per_launch, lo=lo, hi=hi, seed=reads_seed,
                      threads=threads) as nl:
        pipe = _Pipeline(net, write, use_var_type_threshold, stats)
        for batch in nl:
            if not cfg.use_q:
                batch.qual[:] = 0
            if not cfg.use_strand:
                batch.strand[:] = 0
            pipe.submit(batch)
            done += len(batch)
            if log:
                dt = time.perf_counter() - t0
                log("  submitted %d/%d sites (%.0f sites/s)" % (done, hi - lo, done / max(dt, 1e-9)))
        pipe.drain()
    return done


def select_sites(hdf_path: str, holdout_chromosomes=(), site_limit: int = 0, block: int = 8192) -> np.ndarray:
    """Record indices the reference's test loader would visit, in its order (ascending: ``shuffle=False``, main.py:94).

    * ``holdout_chromosomes`` (``--test_holdout_chromosomes``): ONLY records whose VCF chromosome -- the text before the
      first tab of ``vcfrec`` -- is in the set are tested (``ContextDatasetFromNumpy.update_holdout_chromosomes`` /
      ``process_location``, dl4vc/dataset.py:382-395,459-478, and ``AdjustableDataSampler(reverse_holdout=True)``,
      dataset.py:706-711, main.py:88-92);
    * ``site_limit``: stop after that many visited sites (``--max-test-batches``: the reference breaks when
      ``batch > max_test_batches``, i.e. after ``(max_test_batches + 1) * test_batch_size`` sites, trainer.py:513-515)."""
    with CandidateFile(hdf_path) as src:
        n = len(src)
        if holdout_chromosomes:
            want = set(str(c).encode() for c in holdout_chromosomes)
            hits = []
            for b0 in range(0, n, block):
                col = src.read_field(b0, b0 + block, "vcfrec")
                chrom = [bytes(v).split(b"\t", 1)[0] for v in col]
                hits.append(b0 + np.flatnonzero(np.fromiter((c in want for c in chrom), bool, len(chrom))))
                if site_limit > 0 and sum(len(h) for h in hits) >= site_limit:
                    break
            idx = np.concatenate(hits) if hits else np.zeros(0, np.int64)
        else:
            idx = np.arange(n, dtype=np.int64)
    if site_limit > 0:
        idx = idx[:site_limit]
    return idx.astype(np.int64)


def index_runs(idx: np.ndarray):
    """Ascending indices -> maximal runs ``[(lo, hi), ...]`` of consecutive records."""
    if len(idx) == 0:
        return []
    cut = np.flatnonzero(np.diff(idx) != 1) + 1
    starts = np.concatenate(([0], cut))
    ends = np.concatenate((cut, [len(idx)]))
    return [(int(idx[a]), int(idx[b - 1]) + 1) for a, b in zip(starts, ends)]


def run_shard(net, hdf_path: str, out_path: str, shard_index: int = 0, shard_count: int = 1, native: bool = True,
              holdout_chromosomes=(), site_limit: int = 0, **kw) -> int:
    """Score this rank's contiguous slice of the selected sites into ``out_path`` (records only, no header)."""
    from . import loader
    idx = select_sites(hdf_path, holdout_chromosomes, site_limit)
    lo, hi = shard_range(len(idx), shard_index, shard_count)
    runs = index_runs(idx[lo:hi])
    done = 0
    with open(out_path, "w") as f:
        for a, b in runs:
            if native and loader.available():
                done += score_file_native(net, hdf_path, f.write, a, b, **kw)
            else:
                with CandidateFile(hdf_path) as src:
                    done += score_records(net, src, f.write, a, b, **kw)
    return done
