#!/usr/bin/env python3
"""End-to-end rates of the CLI path on the GPU box, written as one JSON record (profiles/rNN_e2e.json):

  * candidates.hdf -> main.py -> epoch1_*.vcf at 100 reads x 201 bp for --precision fp32 and bf16x3: the scoring loop's own
    clock (HDF5 read + site assembly + forward + '%.8f' VCF text) next to the DEVICE-RESIDENT rate of the same network on the same
    kind of pileups (bench.py --reads 100 --skip-empty-rows: what main.py's forward runs), and the whole process incl. start-up;
  * the native loader alone at 1 / 4 / 16 threads;
  * call_variants.sh from a BAM: simulated 30x reads -> tools/convert_bam_single_reads.py (native encoder) -> main.py ->
    format_vcf -> called_variants.vcf.gz, with the encoder's own rate.

Usage: python tools/e2e_rate.py [n_sites [n_bam_locations [out.json]]]"""
import json, os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from dl4vc_amd import synth, hdf5io, loader
from dl4vc_amd.config import DanConfig
from dl4vc_amd.synth import random_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
n_bam = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
out_json = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "e2e.json")
rec = {"sites": n, "shape": "100 reads x 201 bp, production network, seeded random weights, synthetic pileups (dl4vc_amd/synth.py)"}
td = tempfile.mkdtemp(prefix="e2e_")
base = synth.make_sites(128, reads=100, seed=5)
recs = hdf5io.records_from_sites(synth.tile_sites(base, n))
hdf = os.path.join(td, "candidates.hdf")
t0 = time.perf_counter(); hdf5io.write_candidates(hdf, recs)
print("wrote %d records (%.1f MB on disk) in %.1f s" % (n, os.path.getsize(hdf) / 1e6, time.perf_counter() - t0), flush=True)
rec["native_loader_sites_per_s"] = {}
for threads in (1, 4, 16):
    t0 = time.perf_counter()
    with loader.NativeLoader(hdf, reads=100, batch_sites=1024, threads=threads) as nl:
        m = sum(len(b) for b in nl)
    dt = time.perf_counter() - t0
    rec["native_loader_sites_per_s"][str(threads)] = round(m / dt)
    print("native loader, %2d threads: %d sites in %.2f s = %.0f sites/s" % (threads, m, dt, m / dt), flush=True)
cfg = DanConfig()
ck = os.path.join(td, "ckpt.pth.tar")
torch.save({"state_dict": {"module." + k: torch.from_numpy(v) for k, v in random_state_dict(cfg, seed=1).items()}}, ck)
sample = os.path.join(td, "candidates.vcf"); open(sample, "w").write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tCALLED\n")
MODEL = ["--model-conv-layers", "7", "--model-residual-layer-start", "5", "--model-batchnorm", "--model-use-q-scores", "--model-use-strands",
         "--model-use-reads-ref-var-mask", "--model-highway-single-reads", "--model_concat_hw_reads",
         "--model_pool_combine_dimension", "0", "--model_middle_layer_dilation", "2", "--model_final_layer_dilation", "2",
         "--model-hidden-dropout", "0.1"]
rec["main_py"] = {}
for precision in ("fp32", "bf16x3"):
    cmd = [sys.executable, os.path.join(ROOT, "main.py"), "--test_file", hdf, "--modelload", ck, "--sample_vcf", sample,
           "--save_vcf_records", "--save_vcf_records_file", os.path.join(td, "model_test_%s.vcf" % precision), "--sites-per-launch", "4096",
           "--precision", precision] + MODEL
    t0 = time.perf_counter(); r = subprocess.run(cmd, capture_output=True, text=True); dt = time.perf_counter() - t0
    if r.returncode != 0:
        print(r.stderr[-1500:]); sys.exit(1)
    loop = re.search(r"scoring loop .*: (\d+) sites in ([0-9.]+) s = (\d+) sites/s", r.stdout)
    # the device-resident rate of the same forward on the same kind of pileups (main.py computes empty rows once per site)
    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "100", "--sites", str(n), "--steps", "2", "--warmup", "1",
                        "--precision", str(("fp32", "bf16x3").index(precision)), "--skip-empty-rows", "--no-cpu-baseline", "--no-host-path",
                        "--no-oracle-check"], capture_output=True, text=True)
    line = [l for l in b.stdout.splitlines() if l.startswith("{")]
    dev = json.loads(line[-1])["value"] if line else None
    rec["main_py"][precision] = {"scoring_loop_sites_per_s": int(loop.group(3)) if loop else None, "scoring_loop_s": float(loop.group(2)) if loop else None,
                                 "whole_process_s": round(dt, 2), "whole_process_sites_per_s": round(n / dt),
                                 "device_resident_sites_per_s": dev,
                                 "ratio_loop_to_device_resident": round(int(loop.group(3)) / dev, 4) if loop and dev else None}
    print(precision, rec["main_py"][precision], flush=True)

# ---- call_variants.sh from a BAM: simulated reads at ~30x over n_bam candidate positions
if n_bam > 0:
    from dl4vc_amd.bamio import BamWriter, build_bai, CMATCH, FREVERSE
    rng = np.random.default_rng(33)
    span = 400 * n_bam + 2000
    ref = "".join(rng.choice(list("ACGT"), span))
    out = os.path.join(td, "out"); os.makedirs(out)
    fa = os.path.join(td, "ref.fa")
    open(fa, "w").write(">chr20\n" + "\n".join(ref[i:i + 60] for i in range(0, span, 60)) + "\n")
    positions = [1000 + 400 * i for i in range(n_bam)]
    alts = {p: ("A" if ref[p - 1] != "A" else "C") for p in positions}
    bam = os.path.join(td, "reads.bam")
    n_reads = span * 30 // 150
    starts = np.sort(rng.integers(0, span - 150, n_reads))
    pos_set = np.array(positions)
    t0 = time.perf_counter()
    with BamWriter(bam, [("chr20", span)]) as w:
        for i, s in enumerate(starts):
            s = int(s)
            seq = list(ref[s:s + 150])
            if i % 2 == 0:
                lo = np.searchsorted(pos_set, s + 1); hi = np.searchsorted(pos_set, s + 150, side="right")
                for p in pos_set[lo:hi]:
                    seq[int(p) - 1 - s] = alts[int(p)]
            w.write(0, s, "frag%d" % i, FREVERSE if i % 2 else 0, 60, [(CMATCH, 150)], "".join(seq), [30] * 150)
    build_bai(bam, bam + ".bai")
    print("simulated BAM: %d reads over %d bp in %.1f s" % (n_reads, span, time.perf_counter() - t0), flush=True)
    open(os.path.join(out, "candidates.vcf"), "w").write(
        "##fileformat=VCFv4.2\n##contig=<ID=chr20,length=%d>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tCALLED\n" % span +
        "".join("chr20\t%d\t.\t%s\t%s\t50\t.\tDP=30;AF=0.5\tGT\t0/1\n" % (p, ref[p - 1], alts[p]) for p in positions))
    t0 = time.perf_counter()
    r = subprocess.run(["bash", os.path.join(ROOT, "call_variants.sh"), "-m", ck, "-o", out, "-i", bam, "-r", fa, "-p", "16"], capture_output=True, text=True)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        print(r.stdout[-800:], r.stderr[-800:]); sys.exit(1)
    log = open(os.path.join(out, "training_data.log")).read()
    enc = re.search(r"([0-9.]+) ms per location", log)
    t_hdf = os.path.getmtime(os.path.join(out, "candidates.hdf")) - os.path.getmtime(os.path.join(out, "candidates.vcf"))
    rec["call_variants_sh_from_bam"] = {"locations": n_bam, "coverage": "30x, 150-bp reads", "whole_pipeline_s": round(dt, 2),
                                        "locations_per_s": round(n_bam / dt, 1), "encoder_log_tail": log.strip().splitlines()[-1][:200] if log.strip() else "",
                                        "encoder_ms_per_location": float(enc.group(1)) if enc else None,
                                        "seconds_until_candidates_hdf": round(t_hdf, 2),
                                        "called_variants": os.path.isfile(os.path.join(out, "called_variants.vcf.gz"))}
    print(rec["call_variants_sh_from_bam"], flush=True)
os.makedirs(os.path.dirname(out_json), exist_ok=True)
json.dump(rec, open(out_json, "w"), indent=1)
print("wrote", out_json)
