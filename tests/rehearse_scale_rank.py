#!/usr/bin/env python3
"""CPU rehearsal of a multi-GPU scale run (tools/scale_run.sh --rehearse-cpu N): the launch paths and the line schema of
bench.py / main.py with the CPU oracle standing in for the HIP forward (test infrastructure: this file lives under tests/ because
it imports oracle/), everything else the product's own code over gloo -- bench.resolve_ranks / count_ranks, dl4vc_amd.shard,
dl4vc_amd.inference.run_shard, dl4vc_amd.train.GradientExchange.  It proves that N ranks start, are counted, shard the sites
without a collective, exchange gradient buckets, and that the table of tools/scale_table.py reads what comes out -- not a rate.

  --mode infer | train      one rank of a torch.distributed.run launch (or the only one): prints the bench line on rank 0
  --mode cli                the launcher side of `main.py --gpus N`: N shard processes, host-side concat, main.py's log lines"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
from dl4vc_amd.capi import tree_source_hash  # noqa: E402  (needs no built library: the digest of the sources in the tree)


def small():
    from dl4vc_amd.config import DanConfig
    return DanConfig(reads=8, c_init=16, c_final=16, bottleneck=4, fc_sizes=(16, 8))


def init(args):
    import bench
    rank, local_rank, world = bench.resolve_ranks(args, sys.argv[1:])
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    return rank, world, dist, bench.count_ranks(dist, world, "cpu")


def infer(args):
    from dl4vc_amd import synth
    from dl4vc_amd.shard import shard_range
    from oracle.dan_oracle import dan_forward_oracle, random_state_dict
    import torch
    torch.set_num_threads(1)
    rank, world, dist, seen = init(args)
    cfg = small()
    sd = random_state_dict(cfg, seed=2)
    per_rank = 24
    batch = synth.make_sites(per_rank * world, reads=cfg.reads, seed=3)      # the whole job's sites; this rank scores its contiguous shard
    lo, hi = shard_range(len(batch), rank, world)
    mine = batch.slice(lo, hi)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = dan_forward_oracle(sd, cfg, *mine.arrays())
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert out["vt_prob"].shape == (hi - lo, 3)
    if rank == 0:
        print(json.dumps({"metric": "REHEARSAL candidate-variants/sec (CPU oracle double, %d reads)" % cfg.reads, "value": round(len(batch) * args.steps / dt, 2),
                          "unit": "candidate-variants/s", "n_gpus": world, "ranks_seen": seen, "steps": args.steps, "warmup": 0,
                          "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "rehearsal", "build": {"source_hash": tree_source_hash()},
                          "config": {"workload": "CPU rehearsal: %d sites per rank" % per_rank}}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def train(args):
    from dl4vc_amd.train import GradientExchange
    import torch
    torch.set_num_threads(1)
    rank, world, dist, seen = init(args)
    n0, n1 = 759_301, 18_997                                         # bucket sizes with remainders against every world size here
    grad = torch.from_numpy(np.random.default_rng(rank).standard_normal(n0 + n1).astype(np.float32))
    want = None
    if dist is not None:                                             # the mean every rank must end up with, bit for bit
        want = grad.clone()
        dist.all_reduce(want)
        want /= world
    ex = GradientExchange(dist, world, direct=False) if world > 1 else None
    t_ex = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g = grad.clone()
        time.sleep(0.01)                                             # (the device step's place)
        if ex is not None:
            ta = time.perf_counter()
            ex.start(g[n1:]); ex.start(g[:n1]); ex.finish()
            t_ex += time.perf_counter() - ta
            assert torch.allclose(g, want, atol=1e-6), "the exchanged gradient is not the mean over ranks"
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "REHEARSAL training sites/sec (no device step: gradient exchange over gloo)", "value": round(10 * world * args.steps / dt, 2),
                          "unit": "sites/s", "n_gpus": world, "ranks_seen": seen, "steps": args.steps, "warmup": 0, "ms_per_step": round(dt / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "rehearsal",
                          "config": {"workload": "CPU rehearsal"}, "build": {"source_hash": tree_source_hash()},
                          "exchange": None if world == 1 else {"form": "all-reduce", "backend": "gloo", "bucket_floats": [n0, n1],
                                                                "exposed_ms_per_step": round(t_ex / args.steps * 1e3, 3), "normalisers_ms_per_step": 0.0}}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cli_shard(args):
    """One shard process of the rehearsed `main.py --gpus N`: scores its slice with the oracle double, writes the part file and the
    statistics side file main.py's launcher reads."""
    from test_cli_plumbing import OracleNet
    from dl4vc_amd.inference import run_shard
    from dl4vc_amd.shard import part_path
    from oracle.dan_oracle import random_state_dict
    t_proc = time.time()
    i, n = [int(v) for v in args.shard.split("/")]
    net = OracleNet(small(), random_state_dict(small(), seed=2))
    target = part_path(args.out_final, i)
    t0 = time.time()
    done = run_shard(net, args.hdf, target, i, n, sites_per_launch=64, native=False)
    json.dump({"sites": done, "loop_s": time.time() - t0, "process_s": time.time() - t_proc}, open(target + ".stats.json", "w"))


def cli(args):
    from dl4vc_amd import synth, hdf5io
    from dl4vc_amd.shard import part_path, concat_parts
    N = args.gpus
    os.makedirs(args.out, exist_ok=True)
    hdf = os.path.join(args.out, "rehearsal_candidates.hdf")
    sites = 40 * N + 3                                               # a count N does not divide
    hdf5io.write_candidates(hdf, hdf5io.records_from_sites(synth.make_sites(sites, reads=8, seed=21), store_reads=200))
    out_final = os.path.join(args.out, "epoch1_rehearsal.vcf")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--mode", "cli-shard", "--shard", "%d/%d" % (g, N), "--hdf", hdf,
                               "--out-final", out_final]) for g in range(N)]
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise SystemExit("shard process failed: %s" % rcs)
    t_shards = time.time() - t0
    total = 0
    for g in range(N):                                               # main.py's own log lines (main.py:306-328), so that the table reads them the same way
        st = json.load(open(part_path(out_final, g) + ".stats.json"))
        os.remove(part_path(out_final, g) + ".stats.json")
        total += st["sites"]
        print("\tshard %d/%d on device %s: %d sites, scoring loop %.2f s = %.0f sites/s (process %.2f s incl. start-up and "
              "checkpoint load)" % (g, N, "cpu", st["sites"], st["loop_s"], st["sites"] / max(st["loop_s"], 1e-9), st["process_s"]))
    t1 = time.time()
    concat_parts(out_final, N)
    t_cat = time.time() - t1
    assert total == sites and len(open(out_final).read().splitlines()) == sites
    print("\t%d shards: %d sites in %.2f s = %.0f sites/s whole job; host-side concat %.3f s" % (N, total, t_shards + t_cat, total / max(t_shards + t_cat, 1e-9), t_cat))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", required=True)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "scale"))
    ap.add_argument("--shard", default="")
    ap.add_argument("--hdf", default="")
    ap.add_argument("--out-final", default="")
    a = ap.parse_args()
    {"infer": infer, "train": train, "cli": cli, "cli-shard": cli_shard}[a.mode](a)
