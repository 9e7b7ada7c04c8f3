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
from .vcf import scored_record


def score_records(net, source: CandidateFile, write: Callable[[str], None], lo: int = 0, hi: Optional[int] = None,
                  sites_per_launch: int = 4096, reads_seed: int = 0, use_var_type_threshold: bool = False,
                  max_batches: int = 0, log=None) -> int:
    """Score records ``[lo, hi)`` of ``source`` with ``net`` (anything with ``forward_u8`` and ``config``) and hand
    each scored VCF line (with '\\n') to ``write``.  Returns the number of sites scored."""
    cfg = net.config
    hi = len(source) if hi is None else min(hi, len(source))
    done, batches = 0, 0
    t0 = time.perf_counter()
    for b0 in range(lo, hi, sites_per_launch):
        if max_batches > 0 and batches > max_batches:          # trainer.py:513-515 (same off-by-one)
            break
        recs = source.read(b0, min(b0 + sites_per_launch, hi))
        # the seed is tied to the ABSOLUTE record index, so shard boundaries never change a site's read subset
        batch = assemble_batch(recs, cfg.reads, seed=reads_seed + b0, use_q=cfg.use_q, use_strand=cfg.use_strand)
        out = net.forward_u8(*batch.arrays())
        vt = out["vt_prob"]
        bp = (1.0 - vt[:, 0]) if use_var_type_threshold else out["bp"]      # trainer.py:611-621
        write("".join(scored_record(r, b, v) + "\n" for r, b, v in zip(batch.vcfrec, bp, vt)))
        done += len(recs)
        batches += 1
        if log:
            dt = time.perf_counter() - t0
            log("  scored %d/%d sites (%.0f sites/s)" % (done, hi - lo, done / max(dt, 1e-9)))
    return done


def score_file_native(net, hdf_path: str, write: Callable[[str], None], lo: int, hi: int, sites_per_launch: int = 4096,
                      reads_seed: int = 0, use_var_type_threshold: bool = False, max_batches: int = 0, log=None,
                      threads: int = 0) -> int:
    """Same contract as ``score_records`` but fed by the native batched loader (dl4vc_amd/loader.py): chunk
    inflate and site assembly of batch k+1.. run in C++ threads while the GPU scores batch k."""
    import os
    from .loader import NativeLoader
    cfg = net.config
    threads = threads or max(2, min(16, (os.cpu_count() or 4)))
    done, batches = 0, 0
    t0 = time.perf_counter()
    with NativeLoader(hdf_path, cfg.reads, batch_sites=sites_per_launch, lo=lo, hi=hi, seed=reads_seed,
                      threads=threads) as nl:
        for batch in nl:
            if max_batches > 0 and batches > max_batches:
                break
            if not cfg.use_q:
                batch.qual[:] = 0
            if not cfg.use_strand:
                batch.strand[:] = 0
            out = net.forward_u8(*batch.arrays())
            vt = out["vt_prob"]
            bp = (1.0 - vt[:, 0]) if use_var_type_threshold else out["bp"]
            write("".join(scored_record(r, b, v) + "\n" for r, b, v in zip(batch.vcfrec, bp, vt)))
            done += len(batch)
            batches += 1
            if log:
                dt = time.perf_counter() - t0
                log("  scored %d/%d sites (%.0f sites/s)" % (done, hi - lo, done / max(dt, 1e-9)))
    return done


def run_shard(net, hdf_path: str, out_path: str, shard_index: int = 0, shard_count: int = 1, native: bool = True,
              **kw) -> int:
    """Score this rank's contiguous slice into ``out_path`` (records only, no header)."""
    from . import loader
    with CandidateFile(hdf_path) as src:
        n = len(src)
    lo, hi = shard_range(n, shard_index, shard_count)
    with open(out_path, "w") as f:
        if native and loader.available():
            return score_file_native(net, hdf_path, f.write, lo, hi, **kw)
        with CandidateFile(hdf_path) as src:
            return score_records(net, src, f.write, lo, hi, **kw)
