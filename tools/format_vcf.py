#!/usr/bin/env python3
"""Command-line twin of the reference's tools/format_vcf.py (flags of its ``main``, :224-246): scores ->
genotypes.  The logic lives in dl4vc_amd/vcf.py."""
import argparse
import dataclasses
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl4vc_amd.vcf import FormatOptions, format_vcf      # noqa: E402


def main():
    p = argparse.ArgumentParser(description="threshold scored VCF into genotype calls")
    p.add_argument("--input_file", type=str, default="")
    p.add_argument("--output_file", type=str, default="")
    for f in dataclasses.fields(FormatOptions):
        p.add_argument("--" + f.name, type=float, default=f.default)
    p.add_argument("--debug", action="store_true", default=False)
    a = p.parse_args()
    print(a)
    format_vcf(a.input_file, a.output_file, FormatOptions(**{f.name: getattr(a, f.name) for f in dataclasses.fields(FormatOptions)}))


if __name__ == "__main__":
    main()
