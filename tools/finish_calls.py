#!/usr/bin/env python3
"""Multi-allele join + genotype rewrites + bgzip + tabix of the pipeline's last stage without external tools
(reference: call_variants.sh:162-168).  See dl4vc_amd/vcfpost.py for the rules and their parity status (unpinned)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl4vc_amd.vcfpost import finish_calls      # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--input_file", required=True, help="thresholded VCF (output of format_vcf)")
    ap.add_argument("--joined_file", required=True, help="joined, rewritten plain VCF to write")
    ap.add_argument("--output_gz", required=True, help="BGZF-compressed VCF to write (its .tbi index is written next to it)")
    a = ap.parse_args()
    finish_calls(a.input_file, a.joined_file, a.output_gz)
