"""CLI surface, HDF5 reader and the inference harness on CPU (BASELINE config 1: 1k-site plumbing).

The HIP forward cannot run here, so the harness is driven with a test double whose ``forward_u8`` is the
oracle -- test infrastructure only; the product CLI itself refuses to run without the extension + GPU."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from dl4vc_amd import synth, hdf5io, vcf
from dl4vc_amd.config import DanConfig
from dl4vc_amd.inference import score_records, run_shard
from dl4vc_amd.shard import shard_range, parse_shard, concat_parts, part_path
from oracle.dan_oracle import dan_forward_oracle, random_state_dict


class OracleNet:
    """Test double with DanNet's calling surface, backed by the CPU oracle."""
    def __init__(self, cfg, sd):
        self.config, self.sd = cfg, sd

    def forward_u8(self, *planes, aux=False):
        return dan_forward_oracle(self.sd, self.config, *planes)


SMALL = DanConfig(reads=8, c_init=16, c_final=16, bottleneck=4, fc_sizes=(16, 8))


@pytest.fixture(scope="module")
def hdf_1k(tmp_path_factory):
    d = tmp_path_factory.mktemp("cfg1")
    base = synth.make_sites(250, reads=8, seed=21)
    batch = synth.tile_sites(base, 1000)
    recs = hdf5io.records_from_sites(batch, store_reads=200)
    # distinct positions so that the later sort is well defined
    for i in range(1000):
        f = batch.vcfrec[i].split("\t")
        f[1] = str(100000 + 7 * i)
        recs[i]["vcfrec"] = "\t".join(f).encode()
    path = str(d / "candidates.hdf")
    hdf5io.write_candidates(path, recs)
    sample = str(d / "candidates.vcf")
    open(sample, "w").write("##fileformat=VCFv4.2\n##source=test\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tCALLED\n")
    return path, sample, recs


def test_cli_flags_match_reference():
    import sys
    sys.path.insert(0, ROOT)
    from arguments import create_arg_parser
    ref = json.load(open(os.path.join(GOLDEN, "cli_flags.json")))
    ours = {a.dest: a for a in create_arg_parser()._actions if a.option_strings}
    for f in ref:
        a = ours.get(f["dest"])
        assert a is not None, f["flags"]
        assert a.option_strings == f["flags"]
        # argparse's type=None parses to str, so an untyped reference flag (--sample_vcf) equals type=str
        assert (getattr(a.type, "__name__", None) or "str") == (f["type"] or ("str" if not f["store_true"] else None) or "str") \
            or f["store_true"], f["flags"]
        assert a.default == f["default"], f["flags"]
        assert a.nargs == f["nargs"] and a.required == f["required"], f["flags"]
        assert (type(a).__name__ == "_StoreTrueAction") == f["store_true"]


def test_call_variants_flag_line_parses():
    """Every flag call_variants.sh:101-147 passes must parse and map to the production configuration."""
    import sys
    sys.path.insert(0, ROOT)
    from arguments import create_arg_parser
    line = ("--lr 0.0002 --grad-clip 1.0 --label-smoothing 0.001 --model-hidden-dropout 0.1 --model-batchnorm "
            "--num-data-workers 5 --trust-snp-only --non-snp-train-weight 2.0 --fp-train-weight 0.2 --model-use-q-scores "
            "--model-use-strands --auxillary-loss-weight 1.0 --auxillary-loss-bases-weight 0.01 "
            "--auxillary-loss-allele-weight 0.001 --loss-debug-freq 10000 --aux-keep-candidate-af "
            "--model-use-reads-ref-var-mask --close_match_window 2.0 --focal_loss_alpha 1. --focal_loss_gamma 0.2 "
            "--model-conv-layers 7 --model-residual-layer-start 5 --model-ave-pool-layers 2 --early_loss_weight 0.1 "
            "--model-init-conv-channels 128 --rm_var_reads_rate 0.0 --rm_non_var_reads_rate 0.0 "
            "--close_examples_sample_rate 0.15 --delay_augmentation_epochs 1 --learn_early_loss_weight "
            "--model_pool_combine_dimension 0 --model-final-conv-channels 128 --model-bottleneck-size 32 "
            "--model_final_layer_dilation 2 --model_middle_layer_dilation 2 --model_concat_hw_reads "
            "--model-highway-single-reads --log-interval 1 --model-batchnorm --gpus 1 --test-batch-size 200 "
            "--save_vcf_records --save_vcf_records_file out/model_test.vcf --test_file out/candidates.hdf "
            "--sample_vcf out/candidates.vcf --modelload ckpt.pth.tar").split()
    args = create_arg_parser().parse_args(line)
    assert DanConfig.from_args(args) == DanConfig()           # the only published configuration


def test_hdf5_schema_roundtrip(hdf_1k):
    path, _, recs = hdf_1k
    with hdf5io.CandidateFile(path) as f:
        assert len(f) == 1000 and f.dtype.itemsize == 123965
        got = f.read(990, 2000)
        assert len(got) == 10
        np.testing.assert_array_equal(got.view(np.uint8), recs[990:].view(np.uint8))
        assert len(f.read(5, 5)) == 0


def test_config1_plumbing_end_to_end(hdf_1k, tmp_path):
    """HDF5 -> scored VCF -> sort -> format_vcf: order, %.8f formatting, genotypes."""
    path, sample, recs = hdf_1k
    net = OracleNet(SMALL, random_state_dict(SMALL, seed=2))
    out = vcf.start_scored_vcf(sample, str(tmp_path / "model_test.vcf"))
    assert os.path.basename(out) == "epoch1_model_test.vcf"
    with hdf5io.CandidateFile(path) as src, open(out, "a") as f:
        n = score_records(net, src, f.write, sites_per_launch=300)
    assert n == 1000
    lines = open(out).read().splitlines()
    body = [l for l in lines if not l.startswith("#")]
    assert len(lines) - len(body) == 3 and len(body) == 1000
    # record order is the file order and the ID column carries the four scores with 8 decimals
    for i in (0, 499, 999):
        cols = body[i].split("\t")
        assert cols[1] == str(100000 + 7 * i)
        keys = [kv.split("=")[0] for kv in cols[2].split(";")]
        assert keys == ["BP", "NV", "HV", "OV"] and all(len(kv.split("=")[1].split(".")[1]) == 8 for kv in cols[2].split(";"))
    # batching changes nothing but the CPU oracle's last-bit rounding (torch CPU GEMMs are not batch-invariant;
    # the HIP path is, bit for bit -- tests/test_hip_parity.py::test_chunk_and_batch_boundaries...)
    out2 = str(tmp_path / "again.vcf")
    with hdf5io.CandidateFile(path) as src, open(out2, "w") as f:
        score_records(net, src, f.write, sites_per_launch=1000)
    body2 = open(out2).read().splitlines()
    assert [l.split("\t")[:2] + l.split("\t")[3:] for l in body2] == [l.split("\t")[:2] + l.split("\t")[3:] for l in body]
    sc = lambda l: np.array([float(kv.split("=")[1]) for kv in l.split("\t")[2].split(";")])   # noqa: E731
    assert max(np.abs(sc(a) - sc(b)).max() for a, b in zip(body, body2)) < 1e-5
    # genotype stage on the (already position-sorted) file
    called = vcf.format_vcf_lines([l + "\n" for l in lines], vcf.FormatOptions(**vcf.PIPELINE_OPTIONS))
    gts = [l.rstrip("\n").split("\t")[-1].split(":")[0] for l in called if not l.startswith("#")]
    assert set(gts) <= {"0/1", "1/1"} and len(gts) <= 1000
    # ... and the last five commands of the pipeline in process (join, rewrites, bgzip, tabix), on the same file
    import gzip
    from dl4vc_amd import vcfpost
    thres = tmp_path / "model_test_sorted_thres.vcf"
    thres.write_text("".join(called))
    vcfpost.finish_calls(str(thres), str(tmp_path / "join.vcf"), str(tmp_path / "called_variants.vcf.gz"))
    joined = open(tmp_path / "join.vcf").read()
    assert gzip.open(tmp_path / "called_variants.vcf.gz", "rt").read() == joined
    recs_out = [l for l in joined.splitlines() if not l.startswith("#")]
    assert len(recs_out) == len({(l.split("\t")[0], l.split("\t")[1]) for l in recs_out})        # one record per position
    if recs_out:
        first = recs_out[0].split("\t")
        assert vcfpost.tabix_query(str(tmp_path / "called_variants.vcf.gz"), first[0], int(first[1]), int(first[1])) == [recs_out[0]]


def test_shard_ranges_cover_exactly():
    for n in (0, 1, 7, 1000, 65536):
        for g in (1, 2, 3, 8):
            spans = [shard_range(n, i, g) for i in range(g)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert parse_shard("") == (0, 1) and parse_shard("3/8") == (3, 8)
    with pytest.raises(ValueError):
        parse_shard("8/8")


def _rank_main(rank, world, path, out, port, site_limit=None):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = OracleNet(SMALL, random_state_dict(SMALL, seed=2))
    run_shard(net, path, part_path(out, rank), rank, world, sites_per_launch=100, site_limit=site_limit or 0)
    dist.barrier()                         # the only cross-rank step: parts are complete before the concat
    if rank == 0:
        concat_parts(out, world)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_equals_single_process(hdf_1k, tmp_path):
    """world_size 2 over gloo: sites shard with no data-path collective; host-side concat == 1-process output."""
    import torch.multiprocessing as mp
    path, _, _ = hdf_1k
    single = str(tmp_path / "single.vcf")
    # launches of 100 sites fall on the same boundaries with 1 and 2 shards, so even the CPU oracle's
    # batch-dependent rounding is identical and the files must match byte for byte
    run_shard(OracleNet(SMALL, random_state_dict(SMALL, seed=2)), path, single, 0, 1, sites_per_launch=100)
    multi = str(tmp_path / "multi.vcf")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_rank_main, args=(2, path, multi, port), nprocs=2, join=True)
    assert open(multi).read() == open(single).read()
    assert not os.path.exists(part_path(multi, 0))


def test_eight_rank_sharding_with_a_remainder_equals_single_process(hdf_1k, tmp_path):
    """BASELINE config 3's node: world_size 8 over gloo, 997 sites (8 does not divide them: shards of 124 / 125 sites through
    shard_range's i * N // 8 boundaries), no data-path collective, host-side concat in rank order.  Records come out in the
    single-process order, once each; scores agree with the single-process run (the CPU test double's launches fall on other
    boundaries with 8 shards, so its batch-dependent rounding may differ in the last digits: 1e-6, not byte equality)."""
    import torch.multiprocessing as mp
    path, _, _ = hdf_1k
    single = str(tmp_path / "single.vcf")
    run_shard(OracleNet(SMALL, random_state_dict(SMALL, seed=2)), path, single, 0, 1, sites_per_launch=100, site_limit=997)
    multi = str(tmp_path / "multi.vcf")
    port = 27500 + (os.getpid() % 2000)
    mp.spawn(_rank_main, args=(8, path, multi, port, 997), nprocs=8, join=True)
    a, b = open(single).read().splitlines(), open(multi).read().splitlines()
    assert len(a) == len(b) == 997

    def split(line):
        f = line.split("\t")
        scores = [float(kv.split("=")[1]) for kv in f[2].split(";")]
        return f[:2] + f[3:], scores
    for la, lb in zip(a, b):
        (ka, sa), (kb, sb) = split(la), split(lb)
        assert ka == kb
        assert np.allclose(sa, sb, atol=1e-6), (la, lb)
    assert not any(os.path.exists(part_path(multi, r)) for r in range(8))


def _replica_rank_main(rank, world, port, diverge, out_dir):
    import torch
    import torch.distributed as dist
    from dl4vc_amd.shard import check_replicas_agree
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    state = random_state_dict(SMALL, seed=2)
    for i, k in enumerate(k for k in sorted(state) if k.endswith("running_mean")):
        state[k] = state[k] + np.float32(rank)                  # per replica by design: never part of the check
    if diverge and rank == world - 1:
        k = sorted(k for k in state if k.endswith(".weight"))[3]
        v = state[k].copy()
        v.flat[v.size // 2] = np.nextafter(v.flat[v.size // 2], np.float32(np.inf))   # ONE ulp in one element
        state[k] = v

    def all_reduce_max(vec):
        t = torch.from_numpy(np.asarray(vec, np.float64).copy())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.numpy()
    try:
        check_replicas_agree(state, all_reduce_max)
        verdict = "agree"
    except RuntimeError as e:
        verdict = "diverged" if "diverged" in str(e) else "other: %s" % e
    open(os.path.join(out_dir, "rank%d" % rank), "w").write(verdict)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("diverge", [False, True])
def test_sharded_evaluation_checks_that_the_replicas_hold_the_same_parameters(tmp_path, diverge):
    """main.py's sharded evaluation broadcasts only the BatchNorm running statistics and scores every rank's share with that
    rank's own parameters: one all-reduce of a parameter checksum proves they are the same bits -- a one-ulp
    difference in one element on one of three ranks is an error ON EVERY RANK (no rank walks into the evaluation alone)."""
    import torch.multiprocessing as mp
    port = 25500 + (os.getpid() % 2000) + int(diverge)
    mp.spawn(_replica_rank_main, args=(3, port, diverge, str(tmp_path)), nprocs=3, join=True)
    got = [open(str(tmp_path / ("rank%d" % r))).read() for r in range(3)]
    assert got == ["diverged" if diverge else "agree"] * 3, got


def test_shard_arithmetic_at_genome_scale():
    """The partition of a whole-genome candidate set (~4 M sites; the training set of config 4: 77.7 M
    gradient floats exchanged in 1/8 chunks) over 8 ranks: contiguous, exhaustive, sizes differ by at most one, for counts 8
    does not divide."""
    for n in (4_000_003, 77_700_001, 7, 8, 9, 0):
        spans = [shard_range(n, i, 8) for i in range(8)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n


def test_holdout_chromosomes_and_site_limit_select_the_reference_sites(tmp_path):
    """--test_holdout_chromosomes tests ONLY those chromosomes (dataset.py:382-395, :706-711, main.py:88-92);
    --max-test-batches N --test-batch-size S visits (N + 1) * S sites (trainer.py:513-515).  Shards split the SELECTED
    sites; the read-subset seed stays tied to the absolute record index."""
    from dl4vc_amd.inference import select_sites, index_runs
    batch = synth.make_sites(60, reads=8, seed=5)
    recs = hdf5io.records_from_sites(batch, store_reads=200)
    chroms = ["chr1"] * 20 + ["chr20"] * 15 + ["chr2"] * 10 + ["chr20"] * 5 + ["20"] * 10
    for i, c in enumerate(chroms):
        f = batch.vcfrec[i].split("\t")
        f[0], f[1] = c, str(1000 + i)
        recs[i]["vcfrec"] = "\t".join(f).encode()
    path = str(tmp_path / "c.hdf")
    hdf5io.write_candidates(path, recs)
    with hdf5io.CandidateFile(path) as f:
        col = f.read_field(18, 23, "vcfrec")
        assert [bytes(v).split(b"\t")[0] for v in col] == [b"chr1", b"chr1", b"chr20", b"chr20", b"chr20"]
    assert select_sites(path).tolist() == list(range(60))
    sel = select_sites(path, ["chr20"])
    assert sel.tolist() == list(range(20, 35)) + list(range(45, 50))          # exact string match: '20' is another name
    assert index_runs(sel) == [(20, 35), (45, 50)]
    assert select_sites(path, ["chr20", "20"], site_limit=22).tolist() == list(range(20, 35)) + list(range(45, 50)) + [50, 51]
    assert select_sites(path, site_limit=7).tolist() == list(range(7))
    assert index_runs(np.zeros(0, np.int64)) == []
    net = OracleNet(SMALL, random_state_dict(SMALL, seed=2))
    full = str(tmp_path / "full.vcf")
    assert run_shard(net, path, full, sites_per_launch=1000) == 60
    body = open(full).read().splitlines()
    held = str(tmp_path / "held.vcf")
    assert run_shard(net, path, held, holdout_chromosomes=("chr20",), sites_per_launch=1000) == 20
    got = open(held).read().splitlines()
    assert [l.split("\t")[1] for l in got] == [str(1000 + i) for i in sel]
    sc = lambda l: np.array([float(kv.split("=")[1]) for kv in l.split("\t")[2].split(";")])   # noqa: E731
    assert max(np.abs(sc(a) - sc(body[i])).max() for a, i in zip(got, sel)) < 1e-5
    # two shards of the selection, concatenated
    for g in range(2):
        run_shard(net, path, part_path(held, g), g, 2, holdout_chromosomes=("chr20",), sites_per_launch=1000)
    both = open(part_path(held, 0)).read().splitlines() + open(part_path(held, 1)).read().splitlines()
    assert [l.split("\t")[1] for l in both] == [l.split("\t")[1] for l in got]
    lim = str(tmp_path / "lim.vcf")
    assert run_shard(net, path, lim, site_limit=(2 + 1) * 4, sites_per_launch=5) == 12     # --max-test-batches 2 --test-batch-size 4


def test_main_refuses_shuffle_test_and_maps_device_mask(monkeypatch):
    import sys
    sys.path.insert(0, ROOT)
    import main as cli
    with pytest.raises(SystemExit, match="shuffle_test"):
        cli.main(["--test_file", "x.hdf", "--modelload", "c.pt", "--shuffle_test", "--model_pool_combine_dimension", "0"])
    monkeypatch.delenv("DL4VC_FORCE_DEVICE0", raising=False)
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    assert cli.child_devices(3) == ["0", "1", "2"]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5,7")
    assert cli.child_devices(2) == ["4", "5"]
    with pytest.raises(SystemExit, match="lists only 3"):
        cli.child_devices(4)
    monkeypatch.setenv("DL4VC_FORCE_DEVICE0", "1")
    assert cli.child_devices(2) == ["0", "0"]


def test_cli_refuses_without_gpu(hdf_1k, tmp_path):
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    path, sample, _ = hdf_1k
    ck = str(tmp_path / "ckpt.pth.tar")
    torch.save({"epoch": 1, "state_dict": {"module." + k: torch.from_numpy(v) for k, v in random_state_dict(SMALL, seed=2).items()}}, ck)
    cmd = [sys.executable, os.path.join(ROOT, "main.py"), "--test_file", path, "--modelload", ck, "--sample_vcf", sample,
           "--save_vcf_records", "--save_vcf_records_file", str(tmp_path / "model_test.vcf"), "--model-conv-layers", "7",
           "--model-residual-layer-start", "5", "--model-batchnorm", "--model-use-q-scores", "--model-use-strands",
           "--model-use-reads-ref-var-mask", "--model-highway-single-reads", "--model_concat_hw_reads",
           "--model_pool_combine_dimension", "0", "--model_middle_layer_dilation", "2", "--model_final_layer_dilation", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode != 0 and ("no CPU path" in r.stderr or "no HIP device" in r.stderr)


def test_single_layer_needs_equal_widths():
    """model.py:214 vs :257,275: with one conv layer the reference sizes the layer by init_conv_channels and everything behind
    it by final_conv_channels, and fails in its first forward when they differ; here the configuration is refused up front."""
    from dl4vc_amd.config import DanConfig, UnsupportedModelOption
    DanConfig(layers=1, pool_layers=(), residual_start=0, c_init=16, c_final=16)
    with pytest.raises(UnsupportedModelOption, match="single conv layer"):
        DanConfig(layers=1, pool_layers=(), residual_start=0, c_init=16, c_final=48)


def test_bench_rank_count_is_checked_before_anything_runs():
    """bench.py --gpus N: a WORLD_SIZE that is not N, or fewer visible devices than N, ends the run non-zero instead of
    printing a line for a rank count that did not run (the reference's only parallel form is nn.DataParallel over the
    visible devices, main.py:117)."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "4"], capture_output=True, text=True, env=dict(env, WORLD_SIZE="2"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], capture_output=True, text=True, env=dict(env, WORLD_SIZE="1"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()
    for mode in ("infer", "train"):
        # no WORLD_SIZE: the launcher would start the ranks itself -- but only onto devices that exist (none here)
        r = subprocess.run([sys.executable, bench, "--mode", mode, "--gpus", "64"], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode != 0 and "GPU(s) are visible" in r.stderr and not r.stdout.strip()
