"""Training-side batch assembly and example sampling (the training slice of row A2).

Restates what the reference's dataset adds for training on top of the six input planes
(``ContextDatasetFromNumpy._get_generator``, dl4vc/dataset.py:583-680) and its epoch sampler
(``AdjustableDataSampler``, dl4vc/dataset.py:683-749):

* targets per site: ``label`` (the record's field: 0 TP, 1 FN, 2 FP), ``var_type`` / ``is_snp`` / ``var_base_enum`` /
  ``var_ref_enum`` from the VCF line (``utils.parse_vcf``, utils.py:19-72 -- the truth genotype is the optional 11th
  column ``GT:a/b``), ``coverage`` = reads covering the centre column counted from the SELECTED reads when that count is
  positive, else the VCF's DP (dataset.py:603-620), ``allele_freq`` = the candidate's AF (``--aux-keep-candidate-af``, the
  published flag) or the counted variant fraction;
* the example weight ``(is_snp + (1 - is_snp) * non_snp_train_weight)`` (trainer.py:169-172);
* the easy-example sampler: every epoch keeps all examples that are neither "close" (well classified, trainer.py:258-264),
  black-listed nor held out, plus a random ``close_examples_sample_rate`` share of the close ones, in shuffled order.

* the loader workers: ``BatchPrefetcher`` assembles the batches of an epoch ahead of the GPU in ``--num-data-workers``
  processes, each with its own HDF5 handle (the reference: ``DataLoader(num_workers=args.num_data_workers)``, main.py:59-60,
  train_variant_caller.sh:115) -- at 64 sites per 45-ms step one process (≈60 ms per batch of shuffled records) would
  starve the GPU.

Not restated (off in the published scripts, rejected by the CLI when requested): read / reference noise augmentation
(dataset.py:17-80,292-336), dynamic read down-sampling (dataset.py:258-262).
"""
from __future__ import annotations

import collections
from dataclasses import dataclass
from typing import Dict, Iterable, Iterator, List, Optional, Sequence

import numpy as np

from .alleles import parse_candidate, count_center_support
from .dataset import assemble_site
from .synth import SiteBatch

TARGET_KEYS = ("label", "var_type", "allele_freq", "coverage", "var_base_enum", "var_ref_enum", "is_snp", "weight")


@dataclass
class TrainBatch:
    sites: SiteBatch
    targets: Dict[str, np.ndarray]
    index: np.ndarray              # absolute record indices (the loop writes close flags back by index, trainer.py:263)
    blacklist: np.ndarray
    names: List[str]

    def planes(self):
        return self.sites.arrays()

    def __len__(self):
        return len(self.index)


def site_targets(record, site, keep_candidate_af: bool = True) -> Dict[str, float]:
    """dataset.py:584-620 for one assembled site."""
    info = parse_candidate(site.vcfrec)
    coverage = info["coverage"]
    allele_freq = info["allele_freq"]
    cover, _agree, variant = count_center_support(np.ascontiguousarray(site.reads.T), site.ref, info["var_mode"])
    if cover > 0:                                               # dataset.py:614-620
        coverage = cover
        if not keep_candidate_af:
            allele_freq = variant / cover
    return {"label": int(np.asarray(record["label"]).reshape(-1)[0]), "var_type": int(info["var_type"]),
            "allele_freq": float(allele_freq), "coverage": float(coverage), "var_base_enum": int(info["var_base"]),
            "var_ref_enum": int(info["ref_base"]), "is_snp": int(bool(info["is_snp"]))}


def assemble_training_batch(records, indices: Sequence[int], max_reads: int, seed: Optional[int] = None,
                            non_snp_train_weight: float = 1.0, keep_candidate_af: bool = True, use_q: bool = True,
                            use_strand: bool = True, trust_weight=None) -> TrainBatch:
    """``records[i]`` is the structured record of absolute index ``indices[i]``.  ``seed`` pins the read subset of deep
    pileups as in ``dataset.assemble_batch`` (RandomState(seed + absolute index))."""
    sites, tgs = [], []
    for rec, idx in zip(records, indices):
        rng = np.random.RandomState(seed + int(idx)) if seed is not None else None
        site = assemble_site(rec, max_reads, rng, use_q=use_q, use_strand=use_strand)
        sites.append(site)
        tgs.append(site_targets(rec, site, keep_candidate_af))
    stack = lambda f: np.stack([getattr(s, f) for s in sites])   # noqa: E731
    batch = SiteBatch(stack("reads"), stack("qual"), stack("strand"), stack("ref"), stack("ref_mask"), stack("var_mask"),
                      [s.vcfrec for s in sites], np.array([s.num_reads for s in sites], np.int32))
    t = {"label": np.array([g["label"] for g in tgs], np.uint8), "var_type": np.array([g["var_type"] for g in tgs], np.uint8),
         "allele_freq": np.array([g["allele_freq"] for g in tgs], np.float32),
         "coverage": np.array([g["coverage"] for g in tgs], np.float32),
         "var_base_enum": np.array([g["var_base_enum"] for g in tgs], np.uint8),
         "var_ref_enum": np.array([g["var_ref_enum"] for g in tgs], np.uint8),
         "is_snp": np.array([g["is_snp"] for g in tgs], np.uint8)}
    s = t["is_snp"].astype(np.float32)
    w = s + (1.0 - s) * np.float32(non_snp_train_weight)         # trainer.py:169-172
    if trust_weight is not None:
        w = w * np.asarray(trust_weight, np.float32)
    t["weight"] = w.astype(np.float32)
    return TrainBatch(batch, t, np.asarray(indices, np.int64), np.array([s.blacklist for s in sites], bool),
                      [s.name for s in sites])


def read_indices(source, indices: np.ndarray) -> np.ndarray:
    """Records at arbitrary (shuffled) indices: sorted, read in runs of consecutive indices, returned in request order."""
    indices = np.asarray(indices, np.int64)
    order = np.argsort(indices, kind="stable")
    srt = indices[order]
    out = np.empty(len(indices), dtype=source.dtype)
    i = 0
    while i < len(srt):
        j = i
        while j + 1 < len(srt) and srt[j + 1] - srt[j] <= 1:
            j += 1
        block = source.read(int(srt[i]), int(srt[j]) + 1)
        out[order[i:j + 1]] = block[srt[i:j + 1] - srt[i]]
        i = j + 1
    return out


# ---- loader workers ---------------------------------------------------------------------------------------------------
_WORKER_SOURCE = None


def _worker_open(path: str) -> None:
    global _WORKER_SOURCE
    from .hdf5io import CandidateFile
    _WORKER_SOURCE = CandidateFile(path)


def _worker_batch(task):
    indices, kwargs = task
    return assemble_training_batch(read_indices(_WORKER_SOURCE, indices), indices, **kwargs)


class BatchPrefetcher:
    """Training batches assembled ahead of the consumer, in order.

    ``workers > 0``: a pool of that many SPAWNED processes (never forked: the parent has a HIP context), each holding its
    own read-only handle on ``path`` (libhdf5 is not thread-safe, so processes, as the reference's DataLoader workers are);
    at most ``depth`` batches are in flight.  ``workers == 0`` (the reference's "set to 0 if HDF problems"): batches are
    assembled in the calling process, one at a time."""

    def __init__(self, path: str, workers: int = 5, depth: Optional[int] = None):
        self.path, self.workers = path, max(0, int(workers))
        self.depth = int(depth) if depth else 2 * max(self.workers, 1)
        self._pool = None
        self._source = None
        if self.workers > 0:
            import multiprocessing as mp
            self._pool = mp.get_context("spawn").Pool(self.workers, initializer=_worker_open, initargs=(path,))
        else:
            from .hdf5io import CandidateFile
            self._source = CandidateFile(path)

    def batches(self, index_lists: Iterable[Sequence[int]], **kwargs) -> Iterator[TrainBatch]:
        """``assemble_training_batch(records[idx], idx, **kwargs)`` for every index list, yielded in the order given."""
        if self._pool is None:
            for idx in index_lists:
                idx = np.asarray(idx, np.int64)
                yield assemble_training_batch(read_indices(self._source, idx), idx, **kwargs)
            return
        pending = collections.deque()
        it = iter(index_lists)
        done = False
        while True:
            while not done and len(pending) < self.depth:
                try:
                    idx = np.asarray(next(it), np.int64)
                except StopIteration:
                    done = True
                    break
                pending.append(self._pool.apply_async(_worker_batch, ((idx, kwargs),)))
            if not pending:
                return
            yield pending.popleft().get()

    def close(self) -> None:
        if self._pool is not None:
            self._pool.terminate()
            self._pool.join()
            self._pool = None
        if self._source is not None:
            self._source.close()
            self._source = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class EasyExampleSampler:
    """``AdjustableDataSampler`` (dl4vc/dataset.py:683-749, built at main.py:72-74 when ``close_examples_sample_rate < 1``).

    Holds the per-example tables the training loop updates (``close``: trainer.py:263-264 via ``update_close_example``;
    ``blacklist``: trainer.py:267) and yields one epoch's index order.  ``rng`` is a ``np.random.RandomState``: the reference
    draws from numpy's global legacy generator, so ``RandomState(s)`` here reproduces ``np.random.seed(s)`` there draw for
    draw (one ``permutation`` of the close indices, one ``permutation`` of the merged list)."""

    def __init__(self, n_examples: int, close_keep: float = 0.15, holdout: Optional[np.ndarray] = None,
                 reverse_holdout: bool = False, shuffle: bool = True, rng: Optional[np.random.RandomState] = None,
                 plain: bool = False):
        # plain: main.py:75-77 -- with --close_examples_sample_rate >= 1 the reference builds NO sampler: a plain
        # DataLoader(shuffle=True) over the whole dataset, which filters nothing (not the close examples, not the blacklist
        # and -- a quirk kept, with a warning from main.py -- not the held-out chromosomes either)
        self.plain = bool(plain)
        self.n = int(n_examples)
        self.close_keep = float(close_keep)
        self.close = np.zeros(self.n, bool)
        self.blacklist = np.zeros(self.n, bool)
        self.holdout = np.zeros(self.n, bool) if holdout is None else np.asarray(holdout, bool)
        self.reverse_holdout, self.shuffle = reverse_holdout, shuffle
        self.rng = rng if rng is not None else np.random.RandomState()
        self.epochs = 0
        self.epoch_len = self.n

    def update_close(self, indices, flags) -> None:               # trainer.py:25-40
        self.close[np.asarray(indices, np.int64)] = np.asarray(flags, bool)

    def update_blacklist(self, indices, flags) -> None:           # trainer.py:52-59: only ever sets
        idx = np.asarray(indices, np.int64)[np.asarray(flags, bool)]
        self.blacklist[idx] = True

    def epoch(self) -> np.ndarray:
        self.epochs += 1
        if self.plain:
            self.epoch_len = self.n
            return self.rng.permutation(self.n).astype(np.int64)
        if self.reverse_holdout:                                  # dataset.py:706-711: evaluation on the held-out chromosomes only
            order = np.nonzero(~self.close & ~self.blacklist & self.holdout)[0]
        else:
            keep = np.nonzero(~self.close & ~self.blacklist & ~self.holdout)[0]
            n_close = int(self.close.sum())
            n_take = int(self.close_keep * n_close)               # dataset.py:720
            # dataset.py:727-729, quirk included: with no close example `[-0:]` is the WHOLE index array, so the reference
            # permutes all n indices (and keeps none) -- the draw is reproduced so the next shuffle sees the same state
            close_idx = np.argsort(self.close)[-n_close:]
            take = self.rng.permutation(close_idx)[:n_take]
            order = np.concatenate((keep, take)).astype(np.int64)
        self.epoch_len = len(order)
        if self.shuffle:
            return self.rng.permutation(order)                    # dataset.py:744
        return order

    def __len__(self):
        return self.epoch_len
