#!/usr/bin/env python3
"""End-to-end rate of the CLI path on the GPU box: candidates.hdf -> main.py -> epoch1_*.vcf, and of the
native loader alone.  Usage: python tools/e2e_rate.py [n_sites]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from dl4vc_amd import synth, hdf5io, loader
from dl4vc_amd.config import DanConfig
from dl4vc_amd.synth import random_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
td = tempfile.mkdtemp(prefix="e2e_")
base = synth.make_sites(128, reads=100, seed=5)
recs = hdf5io.records_from_sites(synth.tile_sites(base, n))
hdf = os.path.join(td, "candidates.hdf")
t0 = time.perf_counter(); hdf5io.write_candidates(hdf, recs); print("wrote %d records (%.1f MB on disk) in %.1f s" % (n, os.path.getsize(hdf) / 1e6, time.perf_counter() - t0))
for threads in (1, 4, 16):
    t0 = time.perf_counter()
    with loader.NativeLoader(hdf, reads=100, batch_sites=1024, threads=threads) as nl:
        m = sum(len(b) for b in nl)
    dt = time.perf_counter() - t0
    print("native loader, %2d threads: %d sites in %.2f s = %.0f sites/s" % (threads, m, dt, m / dt))
cfg = DanConfig()
ck = os.path.join(td, "ckpt.pth.tar")
torch.save({"state_dict": {"module." + k: torch.from_numpy(v) for k, v in random_state_dict(cfg, seed=1).items()}}, ck)
sample = os.path.join(td, "candidates.vcf"); open(sample, "w").write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tCALLED\n")
cmd = [sys.executable, os.path.join(ROOT, "main.py"), "--test_file", hdf, "--modelload", ck, "--sample_vcf", sample,
       "--save_vcf_records", "--save_vcf_records_file", os.path.join(td, "model_test.vcf"), "--model-conv-layers", "7",
       "--model-residual-layer-start", "5", "--model-batchnorm", "--model-use-q-scores", "--model-use-strands",
       "--model-use-reads-ref-var-mask", "--model-highway-single-reads", "--model_concat_hw_reads",
       "--model_pool_combine_dimension", "0", "--model_middle_layer_dilation", "2", "--model_final_layer_dilation", "2",
       "--model-hidden-dropout", "0.1", "--sites-per-launch", "4096"]
t0 = time.perf_counter(); r = subprocess.run(cmd, capture_output=True, text=True); dt = time.perf_counter() - t0
print([l for l in r.stdout.strip().splitlines() if "scoring loop" in l or "Time elapsed" in l] if r.returncode == 0 else r.stderr[-1500:])
print("main.py end to end (process start, checkpoint load, HDF5 -> scored VCF): %d sites in %.1f s = %.0f sites/s" % (n, dt, n / dt))
