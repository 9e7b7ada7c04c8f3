"""Site sharding across GPUs: contiguous index ranges, no collective.

Candidate sites are independent, so the record range ``[0, N)`` is cut into G contiguous slices
``[g*N//G, (g+1)*N//G)``; each rank (one process per GPU) scores its slice into a part file and the
parts are concatenated in rank order, which reproduces the single-process file byte for byte because
the reference iterates in record order (``shuffle=False``, main.py:94).
"""
from __future__ import annotations

import os
import shutil
from typing import List, Tuple


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
