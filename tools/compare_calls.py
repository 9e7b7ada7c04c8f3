#!/usr/bin/env python3
"""Acceptance tool for the real-data run that cannot happen offline (north_star: scores within 1e-4 of the reference, genotype calls
identical on HG002 chr20):

    tools/compare_calls.py OURS.vcf REFERENCE.vcf [--candidates candidates.hdf] [--json report.json] [format_vcf threshold flags]

Both files are scored VCFs as dl4vc/utils.py:146-178 writes them (``epoch1_<name>.vcf``).  Prints the comparison of
dl4vc_amd/compare.py and exits 0 only if every judged site agrees: scores within --tol, genotype lines (tools/format_vcf.py:92-221 with
call_variants.sh:154-160's thresholds unless flags say otherwise) identical except on sites within --tol of a decision threshold.
With --candidates, sites with more than 100 reads are set apart (the reference scores a random read subset there:
dl4vc/dataset.py:271-281).  INTEGRATION.md section "Acceptance on real data" has the two commands that produce the files."""
import argparse
import dataclasses
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl4vc_amd.vcf import FormatOptions, PIPELINE_OPTIONS                          # noqa: E402
from dl4vc_amd.compare import compare_scored_vcfs, read_num_reads, summary         # noqa: E402


def main(argv=None):
    p = argparse.ArgumentParser(description="compare two scored VCFs: scores, genotype calls, knife-edge attribution")
    p.add_argument("a")
    p.add_argument("b")
    p.add_argument("--candidates", default=None, help="candidates.hdf the two runs scored (for num_reads)")
    p.add_argument("--max-reads", type=int, default=100)
    p.add_argument("--tol", type=float, default=1e-4)
    p.add_argument("--json", default=None)
    for f in dataclasses.fields(FormatOptions):
        p.add_argument("--" + f.name, type=float, default=PIPELINE_OPTIONS.get(f.name, f.default))
    a = p.parse_args(argv)
    opts = FormatOptions(**{f.name: getattr(a, f.name) for f in dataclasses.fields(FormatOptions)})
    nr = read_num_reads(a.candidates) if a.candidates else None
    rep = compare_scored_vcfs(open(a.a).readlines(), open(a.b).readlines(), opts, tol=a.tol, num_reads=nr, max_reads=a.max_reads)
    print("\n".join(summary(rep)))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(rep, f, indent=1)
    return 0 if rep["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
