#!/usr/bin/env python3
"""CLI twin of the reference's ``tools/convert_bam_single_reads.py``: BAM + FASTA + candidate
VCF(s) -> ``candidates.hdf`` in the record layout main.py reads.  Same flags as the reference for the path call_variants.sh
(:87-98) and the training-data recipe (docs/Data.md) take:

    python tools/convert_bam_single_reads.py --input X.bam --fasta-input ref.fa --fp_vcf OUT/candidates.vcf \
        --output OUT/candidates.hdf --max-reads 200 --num-processes 16 --locations-process-step 100000 \
        --max-insert-length 10 --max-insert-length-variant 50 --save-q-scores --save-strand

(``--tp_vcf`` / ``--tp_full_vcf`` / ``--fn_vcf`` label locations 0 / 1 as the reference does.)  Options of the reference that
this path never used are refused, not ignored.  BAM / BAI / FASTA are read by dl4vc_amd/bamio.py (no htslib needed); records
are written in input order; ``--num-processes`` worker processes each take contiguous runs of locations.
"""
import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np                                             # noqa: E402

from dl4vc_amd import hdf5io                                   # noqa: E402
from dl4vc_amd.pileup_encoder import EncoderOptions, encode_locations, locations_from_vcf      # noqa: E402


def _work(task):
    bam, fasta, locs, opt = task
    return encode_locations(bam, fasta, locs, opt)


def main(argv=None):
    ap = argparse.ArgumentParser(description="BAM file to candidate records for single reads")
    ap.add_argument("--input", type=str, required=True, help="input BAM (coordinate-sorted; a .bai beside it is used when present)")
    ap.add_argument("--chrom", type=str, default="22")
    ap.add_argument("--locations", type=str, default=None)
    ap.add_argument("--tp_vcf", type=str, default=None)
    ap.add_argument("--tp_full_vcf", type=str, default=None)
    ap.add_argument("--fp_vcf", type=str, default=None)
    ap.add_argument("--fn_vcf", type=str, default=None)
    ap.add_argument("--restrict_locations_file", type=str, default="")
    ap.add_argument("--restrict_locations", action="store_true", default=False)
    ap.add_argument("--non_restrict_match_random", action="store_true", default=False)
    ap.add_argument("--output", type=str, default="result")
    ap.add_argument("--locations-process-step", type=int, default=100000)
    ap.add_argument("--locations-restart-pos", type=int, default=0)
    ap.add_argument("--locations-append-data", action="store_true", default=False)
    ap.add_argument("--debug", action="store_true")
    ap.add_argument("--min-base-quality", type=int, default=0)
    ap.add_argument("--fasta-input", type=str, default="hs37d5.fa")
    ap.add_argument("--num-processes", type=int, default=10)
    ap.add_argument("--max-loc", type=int, default=0)
    ap.add_argument("--max-reads", type=int, default=1000)
    ap.add_argument("--window-size", type=int, default=100)
    ap.add_argument("--max-insert-length", type=int, default=10)
    ap.add_argument("--max-insert-length-variant", type=int, default=50)
    ap.add_argument("--save-q-scores", action="store_true")
    ap.add_argument("--save-strand", action="store_true")
    args = ap.parse_args(argv)
    for flag, why in (("locations", "numpy location tables"), ("restrict_locations", "location restriction files"),
                      ("non_restrict_match_random", "location restriction files")):
        if getattr(args, flag):
            raise SystemExit("--%s (%s) is not supported by this converter" % (flag, why))
    # convert_bam_single_reads.py:700-701: the record layout always holds both planes
    assert args.save_q_scores, "Too many options, need to run with Q scores"
    assert args.save_strand, "Too many options, need to run with strand save"
    opt = EncoderOptions(window_size=args.window_size, max_reads=args.max_reads, max_insert_length=args.max_insert_length,
                         max_insert_length_variant=args.max_insert_length_variant, min_base_quality=args.min_base_quality)
    locations = []
    if args.tp_vcf:
        locations.extend(locations_from_vcf(args.tp_vcf, label=0, full_vcf=args.tp_full_vcf))
    if args.fn_vcf:
        locations.extend(locations_from_vcf(args.fn_vcf, label=1))
    if args.fp_vcf:
        locations.extend(locations_from_vcf(args.fp_vcf, label=2))
    print("After adding from VCF, %d total locations considered" % len(locations))
    if args.max_loc > 0:
        locations = locations[:args.max_loc]
    n_loc = len(locations)
    start = args.locations_restart_pos
    append = start > 0 or args.locations_append_data
    if append:
        assert os.path.isfile(args.output), "Output file must exist for append mode."
    step = args.locations_process_step
    procs = max(1, args.num_processes)
    print("Processing %d locations with %d process" % (n_loc, procs))
    print("Splitting locations into ~%d chunks [%d each] to save memory..." % (math.ceil(max(n_loc - start, 0) / step), step))
    pool = None
    from dl4vc_amd import loader
    native = loader.available() and not os.environ.get("DL4VC_PILEUP_PYTHON")
    if native:
        # the native encoder (libdl4vc_loader.so, pe_*): --num-processes becomes worker THREADS over contiguous runs of locations
        print("native pileup encoder: %d thread(s)" % procs)
    if not native and procs > 1 and n_loc - start > 4 * procs:
        import multiprocessing as mp
        pool = mp.get_context("spawn").Pool(procs)
    total_errors, written, created = 0, 0, append
    t0 = time.time()
    try:
        while start < n_loc or not created:
            chunk = locations[start:start + step]
            if pool is not None and len(chunk) > 4 * procs:
                per = max(8, math.ceil(len(chunk) / (4 * procs)))
                tasks = [(args.input, args.fasta_input, chunk[i:i + per], opt) for i in range(0, len(chunk), per)]
                parts = pool.map(_work, tasks)
                recs = np.concatenate([p[0] for p in parts]) if parts else np.zeros(0)
                errors = sum(p[1] for p in parts)
            else:
                recs, errors = encode_locations(args.input, args.fasta_input, chunk, opt, native=native, threads=procs)
            total_errors += errors
            if not created:
                hdf5io.write_candidates(args.output, recs, chunk=8)
                created = True
            elif len(recs):
                hdf5io.append_candidates(args.output, recs)
            written += len(recs)
            start += step
            print("Total errors %d through %d steps (%d records, %.1f s)" % (total_errors, min(start, n_loc), written, time.time() - t0), flush=True)
    finally:
        if pool is not None:
            pool.terminate()
            pool.join()
    print("Parsing errors in %d / %d locations -- saved to: %s" % (total_errors, n_loc, args.output))
    return 0


if __name__ == "__main__":
    sys.exit(main())
