"""Site sharding across GPUs: contiguous index ranges, no collective.

Candidate sites are independent, so the record range ``[0, N)`` is cut into G contiguous slices
``[g*N//G, (g+1)*N//G)``; each rank (one process per GPU) scores its slice into a part file and the
parts are concatenated in rank order, which reproduces the single-process file byte for byte because
the reference iterates in record order (``shuffle=False``, main.py:94).
"""
from __future__ import annotations

import os
import shutil
import zlib
from typing import List, Tuple

import numpy as np


def shard_range(n: int, index: int, count: int) -> Tuple[int, int]:
    if count < 1 or not (0 <= index < count):
        raise ValueError("shard %d of %d" % (index, count))
    return (index * n) // count, ((index + 1) * n) // count


def parse_shard(text: str) -> Tuple[int, int]:
    """``"i/n"`` -> (i, n); empty -> (0, 1)."""
    if not text:
        return 0, 1
    i, n = text.split("/")
    i, n = int(i), int(n)
    if n < 1 or not (0 <= i < n):
        raise ValueError("bad --shard %r" % text)
    return i, n


def part_path(path: str, index: int) -> str:
    return "%s.part%d" % (path, index)


def concat_parts(path: str, count: int, header_from: str = None, keep_parts: bool = False) -> str:
    """``path`` = header (the '#' lines of ``header_from`` if given) + part0 + part1 + ...  Parts hold records
    only.  Returns ``path``."""
    with open(path, "wb") as out:
        if header_from:
            with open(header_from, "rb") as f:
                for line in f:
                    if not line.startswith(b"#"):
                        break
                    out.write(line)
        for g in range(count):
            with open(part_path(path, g), "rb") as f:
                shutil.copyfileobj(f, out, 1 << 20)
    if not keep_parts:
        for g in range(count):
            os.remove(part_path(path, g))
    return path


# ------------------------------------------------------------------------------------------
# data-parallel replicas: a cheap proof that every rank holds the same parameters
# ------------------------------------------------------------------------------------------
PER_REPLICA_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked")


def replica_checksum(state) -> np.ndarray:
    """Three numbers that are equal on two ranks exactly when their trained parameters are bit-identical (up to a CRC
    collision): CRC-32 of every parameter's bytes chained in key order, split into two 16-bit halves (so that a float64
    all-reduce carries them exactly), and the tensor count.  BatchNorm running statistics are per replica by design
    (nn.DataParallel keeps replica 0's; main.py:117) and do not take part."""
    crc, n = 0, 0
    for key in sorted(state):
        if key.rsplit(".", 1)[-1] in PER_REPLICA_SUFFIXES:
            continue
        crc = zlib.crc32(np.ascontiguousarray(state[key]).tobytes(), crc)
        n += 1
    return np.array([crc >> 16, crc & 0xFFFF, n], np.float64)


def check_replicas_agree(state, all_reduce_max, what: str = "evaluation") -> None:
    """Raises when any rank's parameters differ from another's.  ``all_reduce_max(vec)`` -> element-wise maximum over ranks of a
    float64 vector.  The sharded evaluation scores each rank's share with that rank's parameters: identical by construction
    (every rank applies the same averaged gradient), and this is the check that the construction held -- a rank that
    diverged (a skipped step, a non-deterministic reduction) would otherwise score with other weights without any error."""
    mine = replica_checksum(state)
    both = np.asarray(all_reduce_max(np.concatenate([mine, -mine])), np.float64)
    hi, lo = both[:3], -both[3:]
    if not np.array_equal(hi, lo):
        raise RuntimeError("%s: the ranks hold different parameters (checksum range %s .. %s, this rank %s); the data-parallel "
                           "replicas have diverged" % (what, lo.astype(np.int64).tolist(), hi.astype(np.int64).tolist(),
                                                        mine.astype(np.int64).tolist()))
