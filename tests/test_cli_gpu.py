"""main.py end to end on the GPU: HDF5 in, epoch1_*.vcf out, scores checked against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from dl4vc_amd import synth, hdf5io
from dl4vc_amd.config import DanConfig
from oracle.dan_oracle import dan_forward_oracle, random_state_dict

pytestmark = pytest.mark.gpu

MODEL_FLAGS = ["--model-conv-layers", "7", "--model-residual-layer-start", "5", "--model-batchnorm", "--model-use-q-scores",
               "--model-use-strands", "--model-use-reads-ref-var-mask", "--model-highway-single-reads",
               "--model_concat_hw_reads", "--model_pool_combine_dimension", "0", "--model_middle_layer_dilation", "2",
               "--model_final_layer_dilation", "2", "--model-hidden-dropout", "0.1"]


def _scores(line):
    return np.array([float(kv.split("=")[1]) for kv in line.split("\t")[2].split(";")])


@pytest.mark.parametrize("gpus", [1, 2])
def test_main_py_inference(tmp_path, gpus):
    import torch
    cfg = DanConfig()                                   # production structure, 100 reads
    sd = random_state_dict(cfg, seed=12)
    ck = str(tmp_path / "ckpt.pth.tar")
    torch.save({"epoch": 3, "best_loss": 0.0, "optimizer": {},
                "state_dict": {"module." + k: torch.from_numpy(v) for k, v in sd.items()}}, ck)
    batch = synth.make_sites(24, reads=100, seed=31)
    recs = hdf5io.records_from_sites(batch)
    # two deep pileups (> 100 stored reads): the read subset is pinned by --reads-seed
    rng = np.random.default_rng(0)
    for i in (3, 17):
        recs[i]["num_reads"] = 160
        recs[i]["single_reads"][100:160] = recs[i]["single_reads"][rng.integers(0, 100, 60)]
    hdf = str(tmp_path / "candidates.hdf")
    hdf5io.write_candidates(hdf, recs)
    sample = str(tmp_path / "candidates.vcf")
    open(sample, "w").write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tCALLED\n")
    out = str(tmp_path / "model_test.vcf")
    env = dict(os.environ)
    if gpus == 2:
        # one physical GPU on the test box: both shard processes use device 0
        env["DL4VC_FORCE_DEVICE0"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "main.py"), "--test_file", hdf, "--modelload", ck, "--sample_vcf", sample,
           "--save_vcf_records", "--save_vcf_records_file", out, "--gpus", str(gpus), "--reads-seed", "77",
           "--sites-per-launch", "16"] + MODEL_FLAGS
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    # every scoring process logs how many of its sites sit within 1e-4 of a genotype threshold (the knife-edge count)
    import re
    counts = [(int(a), int(b)) for a, b in re.findall(r"(\d+) of (\d+) sites lie within 1e-4 of a genotype threshold", r.stdout)]
    assert len(counts) == gpus and sum(b for _, b in counts) == 24, r.stdout[-1500:]
    if gpus == 2:
        # the launcher reports every shard's own scoring-loop rate and the cost of the host-side concat (and leaves no side files)
        shard_lines = re.findall(r"shard (\d)/2 on device \S+: (\d+) sites, scoring loop", r.stdout)
        assert sorted(shard_lines) == [("0", "12"), ("1", "12")], r.stdout[-1500:]
        assert re.search(r"2 shards: 24 sites in .* whole job; host-side concat", r.stdout), r.stdout[-1500:]
        assert not [f for f in os.listdir(str(tmp_path)) if f.endswith(".stats.json") or ".part" in f]
    lines = open(str(tmp_path / "epoch1_model_test.vcf")).read().splitlines()
    assert lines[0].startswith("##fileformat") and lines[1].startswith("#CHROM")
    body = lines[2:]
    assert len(body) == 24
    # oracle on the same assembled sites (same pinned subsets: seed + absolute record index)
    from dl4vc_amd.dataset import assemble_batch
    with hdf5io.CandidateFile(hdf) as f:
        full = []
        for b0 in range(0, 24, 16):
            full.append(assemble_batch(f.read(b0, b0 + 16), 100, seed=77 + b0))
    want_vt, want_bp = [], []
    for b in full:
        o = dan_forward_oracle(sd, cfg, *b.arrays())
        want_vt.append(o["vt_prob"]); want_bp.append(o["bp"])
    want_vt, want_bp = np.concatenate(want_vt), np.concatenate(want_bp)
    for i, line in enumerate(body):
        s = _scores(line)
        assert abs(s[0] - want_bp[i]) < 1e-4 and np.abs(s[1:] - want_vt[i]).max() < 1e-4, (i, s, want_bp[i], want_vt[i])
        assert line.split("\t")[1] == batch.vcfrec[i].split("\t")[1]


def test_main_py_loads_a_legacy_format_checkpoint(tmp_path):
    """`--modelload` with a file in the format the PUBLISHED checkpoint is in (docs/Step-by-step.md:14: written by torch 1.2 --
    the legacy non-zip serialisation -- from a DataParallel model: module.-prefixed keys, BatchNorm counters, an Adam `optimizer`
    entry, main.py:194-199): main.py scores with it, and the scored VCF is byte-identical to the run from the same weights saved
    in today's zip format.  tests/test_checkpoint_formats.py checks the loader itself on CPU."""
    import zipfile
    from test_checkpoint_formats import reference_style_checkpoint
    import torch
    cfg = DanConfig(reads=100)
    ck, sd = reference_style_checkpoint(cfg, seed=21)
    batch = synth.make_sites(12, reads=100, seed=41)
    hdf = str(tmp_path / "candidates.hdf")
    hdf5io.write_candidates(hdf, hdf5io.records_from_sites(batch))
    sample = str(tmp_path / "candidates.vcf")
    open(sample, "w").write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tCALLED\n")
    texts = {}
    for fmt in ("legacy", "zip"):
        path = str(tmp_path / ("checkpoint_%s.pth.tar" % fmt))
        torch.save(ck, path, _use_new_zipfile_serialization=(fmt == "zip"))
        assert zipfile.is_zipfile(path) == (fmt == "zip")
        out = str(tmp_path / ("%s_test.vcf" % fmt))
        cmd = [sys.executable, os.path.join(ROOT, "main.py"), "--test_file", hdf, "--modelload", path, "--sample_vcf", sample,
               "--save_vcf_records", "--save_vcf_records_file", out] + MODEL_FLAGS
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        texts[fmt] = open(str(tmp_path / ("epoch1_%s_test.vcf" % fmt))).read()
    assert texts["legacy"] == texts["zip"]
    body = texts["legacy"].splitlines()[2:]
    assert len(body) == 12
    want = dan_forward_oracle(sd, cfg, *batch.arrays())
    for i, line in enumerate(body):
        s = _scores(line)
        assert abs(s[0] - want["bp"][i]) < 1e-4 and np.abs(s[1:] - want["vt_prob"][i]).max() < 1e-4, (i, s)


def _bench_launch(how, bench_args):
    """`driver`: exactly as the driver launches N > 1 (torch.distributed.run, one rank per GPU); `self`: plain
    `python bench.py --gpus 2` with no WORLD_SIZE -- bench.py then starts the ranks itself (resolve_ranks)."""
    import socket
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BENCH_FORCE_DEVICE0="1", BENCH_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if how == "driver":
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")] + bench_args
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)


@pytest.mark.parametrize("how", ["driver", "self"])
def test_bench_two_rank_launch_path(how):
    """The N > 1 branch of bench.py rehearsed on a one-GPU box: both ranks on device 0, gloo instead of RCCL for the
    barrier and the max of the elapsed times (RCCL refuses two ranks on one device).  Sites shard with no data-path
    collective: main.py:117 / SURVEY.md section 8e."""
    import json
    r = _bench_launch(how, ["--gpus", "2", "--sites", "512", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-skip-pass"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["scaling"] == "weak" and rec["steps"] == 1
    assert ("[bench launcher]" in r.stderr) == (how == "self")
    assert rec["config"]["sites_per_gpu"] == 512 and rec["config"]["parallelism"] == "site-shard x2"
    # whole-job aggregate: both ranks' sites over the max-over-ranks time
    assert abs(rec["value"] * rec["ms_per_step"] * 1e-3 - 1024) < 1.0
    assert rec["parity"]["ok"] and rec["parity"]["tiled_identical"] and rec["parity"]["oracle_sites"] == 32
    assert "cpu_baseline" not in rec
    # (the two ranks write to one stderr: their lines may share a line)
    assert r.stderr.count("[bench rank") == 2 and "[bench rank 0/2]" in r.stderr and "[bench rank 1/2]" in r.stderr


TRAIN_FLAGS = ["--lr", "0.0002", "--grad-clip", "1.0", "--epochs", "1", "--log-interval", "1", "--label-smoothing", "0.001",
               "--batch-size", "8", "--test-batch-size", "6", "--trust-snp-only", "--non-snp-train-weight", "2.0", "--fp-train-weight", "0.2",
               "--auxillary-loss-weight", "1.0", "--auxillary-loss-bases-weight", "0.01", "--auxillary-loss-allele-weight", "0.001",
               "--aux-keep-candidate-af", "--close_match_window", "2.0", "--focal_loss_alpha", "1.", "--focal_loss_gamma", "0.2",
               "--close_examples_sample_rate", "0.15", "--model-ave-pool-layers", "2", "--model-init-conv-channels", "128",
               "--model-final-conv-channels", "128", "--model-bottleneck-size", "32"]


def cfg_width(names, state):
    return state["state_dict"]["module.conv2hidden.1.weight"].shape[1]


@pytest.mark.parametrize("gpus", [1, 2])
def test_main_py_training(tmp_path, gpus):
    """main.py --train_file (main.py:151-199): one epoch of two batches, evaluation, checkpoint; the checkpoint then scores the
    same sites identically through the inference-only mode.  gpus = 2: two ranks share device 0 and average their gradients
    over gloo (RCCL refuses two ranks on one device) -- the data-parallel path of BASELINE config 4."""
    import torch
    from dl4vc_amd.synth import make_labelled_records as make_records
    recs = make_records(17, 100, 900)
    # 17 training sites in batches of 8: the last batch holds ONE site, so with two ranks rank 1 sits that step out
    # (it used to raise while the other rank waited in a collective); 16 test sites in batches of 6: with two
    # ranks the evaluation is sharded 1 + 2 batches and the record text concatenated in rank order
    hdf5io.write_candidates(str(tmp_path / "train.hdf"), recs)
    hdf5io.write_candidates(str(tmp_path / "test.hdf"), recs[:16])
    sample = str(tmp_path / "candidates.vcf")
    open(sample, "w").write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tCALLED\n")
    env = dict(os.environ)
    if gpus == 2:
        env.update(DL4VC_FORCE_DEVICE0="1", DL4VC_DIST_BACKEND="gloo")
    ck = str(tmp_path / "model.pth.tar")
    cmd = [sys.executable, os.path.join(ROOT, "main.py"), "--train_file", str(tmp_path / "train.hdf"), "--test_file", str(tmp_path / "test.hdf"),
           "--modelsave", ck, "--sample_vcf", sample, "--save_vcf_records", "--save_vcf_records_file", str(tmp_path / "model_test.vcf"),
           "--gpus", str(gpus)] + MODEL_FLAGS + TRAIN_FLAGS
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Loss:" in r.stdout and "Test set: Average loss:" in r.stdout
    # utils.py:180-186: os.path.splitext("model.pth.tar") = ("model.pth", ".tar")  ->  <base>_epoch<N><ext>, <base>_best<ext>
    state = torch.load(str(tmp_path / "model.pth_epoch1.tar"), map_location="cpu", weights_only=False)
    assert os.path.isfile(str(tmp_path / "model.pth_best.tar"))
    assert state["epoch"] == 1 and np.isfinite(state["best_loss"])
    # 'optimizer' is torch's own Adam state_dict shape (main.py:198): loadable by a tool written against the reference
    od = dict(state["optimizer"])
    names = od.pop("param_names")
    opt = torch.optim.Adam([torch.nn.Parameter(state["state_dict"]["module." + k].clone()) for k in names], lr=1.0)
    opt.load_state_dict(od)
    i = names.index("conv2hidden.1.weight")
    assert od["state"][i]["step"] == 3 and od["state"][i]["exp_avg"].shape == (1024, cfg_width(names, state))
    assert names.index("vt_output_weights") not in od["state"]
    sdk = state["state_dict"]
    assert all(k.startswith("module.") for k in sdk) and "module.conv2hidden.1.weight" in sdk and "module.bn1D_layers.3.running_var" in sdk
    assert all(torch.isfinite(v).all() for v in sdk.values())
    scored = open(str(tmp_path / "epoch1_model_test.vcf")).read().splitlines()
    assert len([l for l in scored if not l.startswith("#")]) == 16
    # the saved checkpoint in inference-only mode reproduces the epoch's evaluation scores
    out2 = tmp_path / "again"
    out2.mkdir()
    cmd = [sys.executable, os.path.join(ROOT, "main.py"), "--test_file", str(tmp_path / "test.hdf"), "--modelload", str(tmp_path / "model.pth_best.tar"),
           "--sample_vcf", sample, "--save_vcf_records", "--save_vcf_records_file", str(out2 / "model_test.vcf"), "--compute-empty-rows"] + MODEL_FLAGS
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    again = open(str(out2 / "epoch1_model_test.vcf")).read().splitlines()
    assert again == scored


@pytest.mark.parametrize("how", ["driver", "self"])
def test_bench_train_two_rank_launch_path(how):
    """bench.py --mode train with two ranks (both on device 0, gloo): the data-parallel branch -- the flat gradient buffer
    averaged per step -- runs before the driver's multi-GPU bench meets it."""
    import json
    r = _bench_launch(how, ["--mode", "train", "--gpus", "2", "--train-batch", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["unit"] == "sites/s" and rec["config"]["sites_per_gpu_per_step"] == 4
    assert abs(rec["value"] * rec["ms_per_step"] * 1e-3 - 8) < 0.01 and np.isfinite(rec["last_step"]["loss"])


def test_call_variants_sh_from_bam_and_candidate_vcf(tmp_path):
    """call_variants.sh -i BAM -r REF with an OUTDIR that holds only candidates.vcf: the pileup encoder (own BAM / FASTA readers)
    makes candidates.hdf, main.py scores it on the GPU, the VCF back end finishes called_variants.vcf.gz; the scores of the
    scored VCF equal the oracle's on the sites assembled from the encoded records."""
    import gzip
    import torch
    from dl4vc_amd.bamio import BamWriter, build_bai, CMATCH, CINS, FREVERSE
    from dl4vc_amd.dataset import assemble_batch
    rng = np.random.default_rng(33)
    ref = "".join(rng.choice(list("ACGT"), 4000))
    out = tmp_path / "out"
    out.mkdir()
    fa = str(tmp_path / "ref.fa")
    open(fa, "w").write(">chr20\n" + "\n".join(ref[i:i + 60] for i in range(0, 4000, 60)) + "\n")
    positions = [700, 1210, 1850, 2400, 3100]
    alts = {p: ("A" if ref[p - 1] != "A" else "C") for p in positions}
    bam = str(tmp_path / "reads.bam")
    starts = np.sort(rng.integers(300, 3500, 700))
    with BamWriter(bam, [("chr20", 4000)]) as w:
        for i, s in enumerate(starts):
            s, n = int(s), 150
            seq, cigar = list(ref[s:s + n]), [(CMATCH, n)]
            for p in positions:
                if s <= p - 1 < s + n and i % 2 == 0:
                    seq[p - 1 - s] = alts[p]
            if s < 1850 - 1 < s + n - 1 and i % 4 == 0:
                k = 1850 - s
                seq, cigar = seq[:k] + list("TT") + seq[k:], [(CMATCH, k), (CINS, 2), (CMATCH, n - k)]
            w.write(0, s, "frag%d" % i, FREVERSE if i % 2 else 0, 60, cigar, "".join(seq), rng.integers(15, 41, len(seq)).tolist())
    build_bai(bam, bam + ".bai")
    open(str(out / "candidates.vcf"), "w").write(
        "##fileformat=VCFv4.2\n##contig=<ID=chr20,length=4000>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tCALLED\n" +
        "".join("chr20\t%d\t.\t%s\t%s\t50\t.\tDP=40;AF=0.5\tGT\t0/1\n" % (p, ref[p - 1], alts[p]) for p in positions))
    cfg = DanConfig()
    sd = random_state_dict(cfg, seed=14)
    ck = str(tmp_path / "ckpt.pth.tar")
    torch.save({"epoch": 1, "best_loss": 0.0, "optimizer": {}, "state_dict": {"module." + k: torch.from_numpy(v) for k, v in sd.items()}}, ck)
    r = subprocess.run(["bash", os.path.join(ROOT, "call_variants.sh"), "-m", ck, "-o", str(out), "-i", bam, "-r", fa, "-p", "2"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:], open(str(out / "training.log")).read()[-1500:] if (out / "training.log").exists() else "")
    with hdf5io.CandidateFile(str(out / "candidates.hdf")) as f:
        recs = f.read(0, len(f))
    assert len(recs) == 5 and [bytes(x).rstrip(b"\x00").decode() for x in recs["name"]] == ["chr20:%d" % p for p in positions]
    scored = [l for l in open(str(out / "epoch1_model_test.vcf")).read().splitlines() if not l.startswith("#")]
    assert len(scored) == 5
    batch = assemble_batch(recs, cfg.reads, seed=0)
    want = dan_forward_oracle(sd, cfg, *batch.arrays())
    got = np.array([_scores(l) for l in scored])
    k = got.shape[1] - 1
    assert np.abs(got[:, 0] - want["bp"].reshape(-1)).max() < 1e-4 and np.abs(got[:, 1:] - want["vt_prob"][:, :k]).max() < 1e-4
    assert gzip.open(str(out / "called_variants.vcf.gz"), "rt").read().startswith("##fileformat")
    assert os.path.isfile(str(out / "called_variants.vcf.gz.tbi"))


def test_direct_exchange_runs_on_rccl_with_the_zero_copy_gradient_buffer(tmp_path):
    """The direct reduce-scatter / all-gather form of GradientExchange through RCCL itself (one rank: every collective is the
    identity, but all of them run -- all_to_all_single, all_gather_into_tensor and the all_reduce of the remainder -- on the
    side stream, on windows of the library's own hipMalloc'ed gradient buffer wrapped zero-copy as a torch tensor)."""
    script = tmp_path / "one_rank.py"
    script.write_text('''
import os, sys
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tests"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", RANK="0", WORLD_SIZE="1")
import numpy as np, torch, torch.distributed as dist
from golden_util import load_train_case
from test_hip_train import cfg_from, hyper_from
from dl4vc_amd.train import DanTrainer, GradientExchange
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
spec, hyper, w, steps, *_ = load_train_case("train_small")
cfg, hp, st = cfg_from(spec), hyper_from(hyper), steps[0]
tr = DanTrainer(cfg, hp, max_batch=6).load_state_dict(w)
tr.backward_begin(st["planes"], st["targets"], dropout_masks=st["masks"])
g = tr.grad_tensor()
(o0, n0), (o1, n1) = tr.grad_buckets()
ex = GradientExchange(dist, 1, direct=True)
ex.world = 1
tr.wait_bucket(0)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    ex._side = side
    ex._mean(g[o0:o0 + n0 - 3])          # a window whose length the (one) rank divides ...
    dist.all_reduce(g[o0 + n0 - 3:o0 + n0])
out = tr.backward_end()
with torch.cuda.stream(side):
    ex._mean(g[o1:o1 + n1])
ex.finish()
after = g.clone()
ref = DanTrainer(cfg, hp, max_batch=6).load_state_dict(w)
ref.backward(st["planes"], st["targets"], dropout_masks=st["masks"])
assert torch.equal(after, ref.grad_tensor()), "the exchange changed a one-rank gradient"
norm = tr.apply()
assert abs(norm - float(st["grad_norm"])) <= 1e-4 * float(st["grad_norm"])
dist.destroy_process_group()
print("ok", out["loss"])
''' % (ROOT, ROOT))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-800:], r.stderr[-2500:])


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_acceptance_procedure_end_to_end_on_synthetic_candidates(tmp_path, precision):
    """INTEGRATION.md section 9, rehearsed: `main.py` scores a candidates.hdf on the GPU (A); the oracle -- standing in for the
    reference's own run, which it is pinned to -- scores the same sites and its scored VCF is written by the same %.8f splice (B);
    `tools/compare_calls.py A B --candidates candidates.hdf` must accept: every score within 1e-4, genotype lines of the two identical
    away from decision thresholds, the two deep pileups (> 100 reads: a random read subset in the reference) set apart."""
    import json
    import torch
    from dl4vc_amd.dataset import assemble_batch
    from dl4vc_amd.vcf import start_scored_vcf, append_scored_records
    cfg = DanConfig()
    sd = random_state_dict(cfg, seed=12)
    ck = str(tmp_path / "ckpt.pth.tar")
    torch.save({"epoch": 3, "best_loss": 0.0, "optimizer": {},
                "state_dict": {"module." + k: torch.from_numpy(v) for k, v in sd.items()}}, ck)
    n = 48
    batch = synth.make_sites(n, reads=100, seed=57)
    recs = hdf5io.records_from_sites(batch)
    rng = np.random.default_rng(1)
    for i in (5, 30):
        recs[i]["num_reads"] = 150
        recs[i]["single_reads"][100:150] = recs[i]["single_reads"][rng.integers(0, 100, 50)]
    hdf = str(tmp_path / "candidates.hdf")
    hdf5io.write_candidates(hdf, recs)
    sample = str(tmp_path / "candidates.vcf")
    open(sample, "w").write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tCALLED\n")
    out = str(tmp_path / "model_test.vcf")
    cmd = [sys.executable, os.path.join(ROOT, "main.py"), "--test_file", hdf, "--modelload", ck, "--sample_vcf", sample,
           "--save_vcf_records", "--save_vcf_records_file", out, "--reads-seed", "91", "--sites-per-launch", "16",
           "--precision", precision] + MODEL_FLAGS
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    ours = str(tmp_path / "epoch1_model_test.vcf")
    # B: the oracle's scores through the same writer (dl4vc/utils.py:146-178's format)
    refdir = tmp_path / "ref"
    refdir.mkdir()
    theirs = start_scored_vcf(sample, str(refdir / "model_test.vcf"))
    with hdf5io.CandidateFile(hdf) as f:
        for b0 in range(0, n, 16):
            b = assemble_batch(f.read(b0, min(n, b0 + 16)), 100, seed=91 + b0)
            o = dan_forward_oracle(sd, cfg, *b.arrays())
            append_scored_records(theirs, o["bp"], o["vt_prob"], b.vcfrec)
    rep = str(tmp_path / "acceptance.json")
    c = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_calls.py"), ours, theirs, "--candidates", hdf, "--json", rep],
                       capture_output=True, text=True)
    print(c.stdout)
    assert c.returncode == 0, c.stdout[-2000:] + c.stderr[-1000:]
    d = json.load(open(rep))
    assert d["ok"] and d["records_a"] == n and d["sites_with_more_reads_than_the_reference_keeps"] == 2 and d["sites_deterministic"] == n - 2
    assert d["genotype_differences"]["elsewhere"]["count"] == 0
    assert all(v["max_abs_diff"] <= 1e-4 for v in d["scores"].values())
