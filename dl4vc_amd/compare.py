"""Acceptance comparison of two scored VCFs -- ours and one the reference wrote (north_star: "softmax scores within 1e-4 of the
reference PyTorch fp32 forward and genotype calls bit-identical on the same HDF5 input").

The real-data run (HG002 chr20; checkpoint and candidates live on S3, docs/Step-by-step.md:14,135) cannot happen offline; this is
the tool that judges it the day both files exist.  Inputs are the files ``dl4vc/utils.py:146-178`` writes (``epoch1_<name>.vcf``:
the candidate records with ``BP=%.8f;NV=%.8f;HV=%.8f;OV=%.8f`` in the ID column, in candidate order).  What is compared:

* the four scores of every record (max |difference|, how many beyond the tolerance);
* the genotype lines ``tools/format_vcf.py:92-221`` derives from each file (``dl4vc_amd.vcf.format_vcf_lines`` on the sorted lines, the
  pipeline's own order: call_variants.sh:151-160) -- whether a record is called at all, its genotype, its quality bucket;
* every difference is attributed: a record whose scores (in either file, or those of another allele at its position -- the
  multi-allele pruning looks at the whole group) lie within the tolerance of a decision threshold is a KNIFE-EDGE site, where two correct
  evaluations may disagree (``dl4vc_amd.vcf.threshold_distance``); a difference anywhere else is a real one;
* with the candidates file: sites with more than ``max_reads`` (100) reads are set apart -- there the reference scores a RANDOM subset
  of the reads (dl4vc/dataset.py:271-281, unseeded in main.py), so its own two runs differ; only sites with ``num_reads <= max_reads``
  are deterministic and judged.

``compare_scored_vcfs`` returns a report dict whose ``"ok"`` is False on any difference elsewhere (scores beyond the tolerance or a
genotype difference away from every threshold, on a deterministic site) or when the two files do not hold the same records."""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Optional, Sequence

import numpy as np

from .vcf import FormatOptions, PIPELINE_OPTIONS, format_vcf_lines, sort_scored_vcf_lines, threshold_distance

SCORE_KEYS = ("BP", "NV", "HV", "OV")


def _records(lines: Sequence[str]):
    """[(key, scores[4], line)] of the data lines, in file order; key = (CHROM, POS, REF, ALT, occurrence)."""
    out = []
    seen: Dict[tuple, int] = defaultdict(int)
    for line in lines:
        if not line.strip() or line.startswith("#"):
            continue
        c = line.rstrip("\n").split("\t")
        if len(c) < 5:
            raise ValueError("not a VCF record: %r" % line[:80])
        try:
            sc = {k: float(v) for k, v in (kv.split("=") for kv in c[2].split(";"))}
            scores = [sc[k] for k in SCORE_KEYS]
        except (ValueError, KeyError):
            raise ValueError("ID column carries no BP/NV/HV/OV scores (not a scored VCF?): %r" % c[2][:60])
        base = (c[0], c[1], c[3], c[4])
        out.append((base + (seen[base],), scores, line if line.endswith("\n") else line + "\n"))
        seen[base] += 1
    return out


def _calls(lines: Sequence[str], options: FormatOptions):
    """{key: (GT, quality bucket)} after the pipeline's sort + format_vcf."""
    out = {}
    seen: Dict[tuple, int] = defaultdict(int)
    for line in format_vcf_lines(sort_scored_vcf_lines(list(lines)), options):
        if line.startswith("#"):
            continue
        c = line.rstrip("\n").split("\t")
        base = (c[0], c[1], c[3], c[4])
        gt, _, q = c[9].partition(":")
        out[base + (seen[base],)] = (gt, q)
        seen[base] += 1
    return out


def compare_scored_vcfs(lines_a: Sequence[str], lines_b: Sequence[str], options: Optional[FormatOptions] = None, tol: float = 1e-4,
                        num_reads: Optional[Sequence[int]] = None, max_reads: int = 100, max_listed: int = 20) -> dict:
    """``lines_a`` / ``lines_b``: the two scored VCFs (header + records, candidate order).  ``num_reads``: per record of A (file order),
    from the candidates file; None = every site is judged."""
    o = options or FormatOptions(**PIPELINE_OPTIONS)
    ra, rb = _records(lines_a), _records(lines_b)
    ka, kb = {k for k, _, _ in ra}, {k for k, _, _ in rb}
    rep: dict = {"records_a": len(ra), "records_b": len(rb), "only_in_a": len(ka - kb), "only_in_b": len(kb - ka), "tolerance": tol,
                 "format_options": dict(o.__dict__)}
    if num_reads is not None and len(num_reads) != len(ra):
        raise ValueError("the candidates file holds %d records, the first VCF %d" % (len(num_reads), len(ra)))
    common = [k for k, _, _ in ra if k in kb]
    sa = {k: s for k, s, _ in ra}
    sb = {k: s for k, s, _ in rb}
    det = {k: True for k in common}
    if num_reads is not None:
        for (k, _, _), n in zip(ra, num_reads):
            if k in det:
                det[k] = int(n) <= max_reads
    rep["sites_deterministic"] = sum(det.values())
    rep["sites_with_more_reads_than_the_reference_keeps"] = len(common) - rep["sites_deterministic"]

    # ---- scores
    A = np.array([sa[k] for k in common], np.float64).reshape(-1, 4)
    B = np.array([sb[k] for k in common], np.float64).reshape(-1, 4)
    D = np.abs(A - B)
    dmask = np.array([det[k] for k in common], bool)
    rep["scores"] = {}
    for j, name in enumerate(SCORE_KEYS):
        col = D[:, j]
        rep["scores"][name] = {"max_abs_diff": float(col[dmask].max(initial=0.0)), "beyond_tolerance": int((col[dmask] > tol).sum()),
                               "max_abs_diff_random_subset_sites": float(col[~dmask].max(initial=0.0))}

    # ---- distance to the nearest decision threshold, per record and per (chrom, pos) group (the multi-allele rules look at the group)
    recs = [l for _, _, l in ra if True]
    keys_a = [k for k, _, _ in ra]
    dist_a = dict(zip(keys_a, threshold_distance([l for l in recs], np.array([[s[1], s[2], s[3]] for _, s, _ in ra]).reshape(-1, 3), o)))
    dist_b = dict(zip([k for k, _, _ in rb],
                      threshold_distance([l for _, _, l in rb], np.array([[s[1], s[2], s[3]] for _, s, _ in rb]).reshape(-1, 3), o)))
    group_dist: Dict[tuple, float] = defaultdict(lambda: float("inf"))
    group_det: Dict[tuple, bool] = defaultdict(lambda: True)
    for k in common:
        g = k[:2]
        group_dist[g] = min(group_dist[g], float(dist_a[k]), float(dist_b[k]))
        group_det[g] = group_det[g] and det[k]
    near = np.array([min(dist_a[k], dist_b[k]) <= tol for k in common], bool)
    rep["sites_within_tolerance_of_a_threshold"] = int((near & dmask).sum())

    # ---- genotype calls
    ca, cb = _calls(lines_a, o), _calls(lines_b, o)
    rep["calls_a"], rep["calls_b"] = len(ca), len(cb)
    diffs = {"knife_edge": [], "elsewhere": [], "random_subset_sites": []}
    bucket_only = 0
    for k in common:
        a, b = ca.get(k), cb.get(k)
        if a == b:
            continue
        if a is not None and b is not None and a[0] == b[0]:
            bucket_only += 1                                   # same call, neighbouring quality bucket (int() of a score difference)
            continue
        g = k[:2]
        entry = {"site": "%s:%s %s>%s" % k[:4], "a": None if a is None else "%s:%s" % a, "b": None if b is None else "%s:%s" % b,
                 "scores_a": sa[k], "scores_b": sb[k], "distance_to_threshold": float(group_dist[g])}
        if not group_det[g]:
            diffs["random_subset_sites"].append(entry)
        elif group_dist[g] <= tol:
            diffs["knife_edge"].append(entry)
        else:
            diffs["elsewhere"].append(entry)
    rep["genotype_differences"] = {k: {"count": len(v), "first": v[:max_listed]} for k, v in diffs.items()}
    rep["quality_bucket_only_differences"] = bucket_only
    rep["calls_identical"] = not any(diffs.values())
    scores_ok = all(v["beyond_tolerance"] == 0 for v in rep["scores"].values())
    rep["ok"] = bool(rep["only_in_a"] == 0 and rep["only_in_b"] == 0 and scores_ok and not diffs["elsewhere"])
    return rep


def read_num_reads(candidates_path: str) -> np.ndarray:
    """``num_reads`` of every record of a candidates.hdf (dl4vc/dataset.py's HDF5 schema), in record order."""
    from .hdf5io import CandidateFile
    with CandidateFile(candidates_path) as f:
        n = len(f)
        out = np.empty(n, np.int64)
        step = 65536
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            out[lo:hi] = np.asarray(f.read_field(lo, hi, "num_reads")).reshape(-1)
    return out


def summary(rep: dict) -> List[str]:
    s = ["records: %d / %d (only in A: %d, only in B: %d); judged (<= max reads): %d, set apart (random read subset in the reference): %d"
         % (rep["records_a"], rep["records_b"], rep["only_in_a"], rep["only_in_b"], rep["sites_deterministic"],
            rep["sites_with_more_reads_than_the_reference_keeps"])]
    for k, v in rep["scores"].items():
        s.append("  %s: max |difference| %.3g, %d beyond %.0e%s" % (k, v["max_abs_diff"], v["beyond_tolerance"], rep["tolerance"],
                 ("  (random-subset sites: %.3g)" % v["max_abs_diff_random_subset_sites"]) if rep["sites_with_more_reads_than_the_reference_keeps"] else ""))
    s.append("sites within %.0e of a format_vcf threshold: %d" % (rep["tolerance"], rep["sites_within_tolerance_of_a_threshold"]))
    g = rep["genotype_differences"]
    s.append("genotype lines: %d / %d; differences: %d on knife-edge sites, %d ELSEWHERE, %d on random-subset sites; %d quality-bucket-only"
             % (rep["calls_a"], rep["calls_b"], g["knife_edge"]["count"], g["elsewhere"]["count"], g["random_subset_sites"]["count"],
                rep["quality_bucket_only_differences"]))
    for kind in ("elsewhere", "knife_edge"):
        for e in g[kind]["first"]:
            s.append("  [%s] %s: %s vs %s (distance to the nearest threshold %.3g)" % (kind, e["site"], e["a"], e["b"], e["distance_to_threshold"]))
    s.append("RESULT: %s" % ("identical within the bars" if rep["ok"] else "DIFFERENT"))
    return s
