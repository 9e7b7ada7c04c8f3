#!/usr/bin/env python3
"""Helper of tools/scale_run.sh: generates the CLI inputs (--make-inputs), prints the model flags (--model-flags) and condenses
the bench lines and the main.py log of a scale run into ONE table (--table)."""
import argparse
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODEL_FLAGS = ["--model-conv-layers", "7", "--model-residual-layer-start", "5", "--model-batchnorm", "--model-use-q-scores",
               "--model-use-strands", "--model-use-reads-ref-var-mask", "--model-highway-single-reads", "--model_concat_hw_reads",
               "--model_pool_combine_dimension", "0", "--model_middle_layer_dilation", "2", "--model_final_layer_dilation", "2",
               "--model-hidden-dropout", "0.1"]


def make_inputs(out, sites):
    """candidates.hdf (``sites`` records: 128 distinct synthetic pileups tiled, written in pieces so that a million records never
    sit in memory at once), a seeded checkpoint in the reference's format, a header-only sample VCF."""
    import numpy as np
    import torch
    from dl4vc_amd import synth, hdf5io
    from dl4vc_amd.config import DanConfig
    from dl4vc_amd.synth import random_state_dict
    base = synth.make_sites(128, reads=100, seed=5)
    piece = hdf5io.records_from_sites(synth.tile_sites(base, 16384))
    path = os.path.join(out, "candidates.hdf")
    done = 0
    while done < sites:
        n = min(len(piece), sites - done)
        if done == 0:
            hdf5io.write_candidates(path, piece[:n])
        else:
            hdf5io.append_candidates(path, piece[:n])
        done += n
    cfg = DanConfig()
    torch.save({"epoch": 1, "best_loss": 0.0, "optimizer": {},
                "state_dict": {"module." + k: torch.from_numpy(v) for k, v in random_state_dict(cfg, seed=1).items()}}, os.path.join(out, "ckpt.pth.tar"))
    open(os.path.join(out, "candidates.vcf"), "w").write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tCALLED\n")
    print("wrote %d records to %s (%.1f MB)" % (sites, path, os.path.getsize(path) / 1e6))


def last_line(path):
    if not os.path.isfile(path):
        return None
    ls = [l for l in open(path) if l.startswith("{")]
    return json.loads(ls[-1]) if ls else None


def table(out, gpus):
    rows = []
    base = {}
    base_lib = {}
    mixed = []
    for mode in ("infer", "train"):
        for f in sorted(glob.glob(os.path.join(out, "%s_n*.json" % mode)), key=lambda p: int(re.search(r"_n(\d+)\.json", p).group(1))):
            n = int(re.search(r"_n(\d+)\.json", f).group(1))
            rec = last_line(f)
            if rec is None:
                rows.append((mode, n, "FAILED (see %s)" % f.replace(".json", ".err"), "", "", "", "", ""))
                continue
            # which library wrote the line (bench.py's `build.source_hash` = dan_source_hash() of the loaded libdl4vc_dan.so): printed
            # beside every row, and a row whose library is not the n = 1 row's says so -- a table can never silently mix builds
            lib = str((rec.get("build") or {}).get("source_hash", "?"))
            if n == 1:
                base[mode] = rec["value"]
                base_lib.setdefault("hash", lib)
            eff = "%.3f" % (rec["value"] / (n * base[mode])) if mode in base else "-"
            if "hash" in base_lib and lib != base_lib["hash"]:
                lib += " MIXED"
                mixed.append((mode, n))
            ex = rec.get("exchange") or {}
            rows.append((mode, n, "%.0f %s" % (rec["value"], rec["unit"]), "%.2f" % rec["ms_per_step"], eff,
                         "%d" % rec.get("ranks_seen", -1), lib,
                         ("%.2f ms exposed + %.2f ms normalisers, %s" % (ex["exposed_ms_per_step"], ex["normalisers_ms_per_step"], ex["form"])) if ex else
                         ("no exchange (one rank)" if mode == "train" else "no collective on the data path")))
    print()
    print("%-6s %3s  %-34s %12s %10s %10s  %-22s %s" % ("mode", "n", "whole-job rate", "ms per step", "efficiency", "ranks_seen", "library (source hash)", "gradient exchange"))
    for r in rows:
        print("%-6s %3d  %-34s %12s %10s %10s  %-22s %s" % r)
    if mixed:
        print("WARNING: rows %s were written by another build of libdl4vc_dan.so than the n = 1 row (%s): efficiencies across them mean nothing"
              % (", ".join("%s n=%d" % m for m in mixed), base_lib["hash"]))
    log = os.path.join(out, "main_n%d.txt" % gpus)
    print()
    if os.path.isfile(log):
        text = open(log).read()
        shards = re.findall(r"shard (\d+)/(\d+) on device (\S+): (\d+) sites, scoring loop ([0-9.]+) s = (\d+) sites/s \(process ([0-9.]+) s", text)
        whole = re.search(r"(\d+) shards: (\d+) sites in ([0-9.]+) s = (\d+) sites/s whole job; host-side concat ([0-9.]+) s", text)
        print("main.py --gpus %d (contiguous shards, one process per GPU, host-side concat):" % gpus)
        for g, n, dev, s, t, rate, proc in shards:
            print("   shard %s/%s on device %s: %s sites, scoring loop %s s = %s sites/s, process %s s" % (g, n, dev, s, t, rate, proc))
        if whole:
            print("   whole job: %s sites in %s s = %s sites/s; host-side concat %s s" % (whole.group(2), whole.group(3), whole.group(4), whole.group(5)))
        elif not shards:
            one = re.search(r"scoring loop .*: (\d+) sites in ([0-9.]+) s = (\d+) sites/s", text)
            print("   one process: %s" % (one.group(0) if one else "no scoring-loop line found in %s" % log))
    else:
        print("no main.py log at %s" % log)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--make-inputs")
    ap.add_argument("--sites", type=int, default=1048576)
    ap.add_argument("--model-flags", action="store_true")
    ap.add_argument("--table")
    ap.add_argument("--gpus", type=int, default=1)
    a = ap.parse_args()
    if a.model_flags:
        print(" ".join(MODEL_FLAGS))
    if a.make_inputs:
        make_inputs(a.make_inputs, a.sites)
    if a.table:
        table(a.table, a.gpus)
