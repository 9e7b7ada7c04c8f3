"""Training-side host logic: batch assembly with targets (dataset.py:583-680) and the easy-example sampler
(dataset.py:683-749), against what the reference produced (tests/golden/train_*.npz hold the batches its own dataset +
DataLoader collated; train_sampler.json its sampler's index orders)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from golden_util import load_train_case, train_cases
from dl4vc_amd.train_data import assemble_training_batch, EasyExampleSampler
from dl4vc_amd.train import TrainHyper, average_gradients


def _records(n_sites, reads, seed):
    """The record array the fixture generator fed to the reference's dataset (oracle/gen_golden_train.py::make_records is
    deterministic; rebuilt here from the same synthetic sites)."""
    from dl4vc_amd.synth import make_labelled_records as make_records
    return make_records(n_sites, reads, seed)


CASE_SEEDS = {"train_small": (6, 2, 300), "train_var_nobn": (4, 1, 400), "train_var_pool24": (4, 1, 410), "train_var_l5res2": (4, 1, 420),
              "train_var_nohw": (4, 1, 430), "train_var_cfinal": (4, 1, 440)}


@pytest.mark.parametrize("case", sorted(CASE_SEEDS))
def test_training_batches_equal_the_reference_datasets(case):
    n_sites, n_steps, seed = CASE_SEEDS[case]
    spec, hyper, _w, steps, *_ = load_train_case(case)
    recs = _records(n_sites * n_steps, spec["reads"], seed)
    hp = TrainHyper(**{k: v for k, v in hyper.items() if k in TrainHyper.__dataclass_fields__})
    for s, st in enumerate(steps):
        idx = np.arange(s * n_sites, (s + 1) * n_sites)
        b = assemble_training_batch(recs[idx], idx, spec["reads"], seed=0, non_snp_train_weight=hp.non_snp_train_weight)
        for got, want, nm in zip(b.planes(), st["planes"], ("reads", "qual", "strand", "ref", "ref_mask", "var_mask")):
            np.testing.assert_array_equal(got, want, err_msg=nm)
        for k, want in st["targets"].items():
            np.testing.assert_allclose(b.targets[k].astype(np.float64), np.asarray(want, np.float64).reshape(-1), rtol=1e-6, err_msg=k)
        assert b.sites.vcfrec == st["vcfrec"] and not b.blacklist.any()


def test_easy_example_sampler_reproduces_the_reference_draw_for_draw():
    for c in json.load(open(os.path.join(GOLDEN, "train_sampler.json"))):
        sm = EasyExampleSampler(c["n"], close_keep=c["close_keep"], holdout=np.array(c["holdout"]), reverse_holdout=c["reverse_holdout"],
                                shuffle=c["shuffle"], rng=np.random.RandomState(c["seed"]))
        sm.close[:] = c["close"]
        sm.blacklist[:] = c["blacklist"]
        assert sm.epoch().tolist() == c["epoch1"], c["name"]
        assert sm.epoch().tolist() == c["epoch2"], c["name"]
        assert len(sm) == c["len"]
    sm = EasyExampleSampler(10, rng=np.random.RandomState(0))
    sm.update_close([1, 3], [True, False])
    sm.update_blacklist([2, 5], [False, True])
    assert sm.close.tolist() == [False, True] + [False] * 8 and sm.blacklist[5] and not sm.blacklist[2]
    assert 5 not in sm.epoch().tolist()


def _rank(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.arange(8, dtype=torch.float32) * (rank + 1)          # this rank's flat gradient buffer
    average_gradients(g, world, dist.all_reduce)
    q.put((rank, g.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gradient_average_world2_gloo():
    """One collective over the flat gradient buffer between backward and apply (replaces nn.DataParallel's reduce-add,
    main.py:117): every rank ends with the mean of the per-rank gradients."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    want = (np.arange(8) * 1.5).tolist()
    assert got[0] == want and got[1] == want
    g = np.ones(3, np.float32)
    average_gradients(g, 1, None)                                   # world 1: untouched, no collective
    assert g.tolist() == [1, 1, 1]


def _rank_exchange(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from dl4vc_amd.train import GradientExchange
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    for direct in (True, False):
        g = (torch.arange(17, dtype=torch.float32) + 1) * (rank + 1) ** 2      # this rank's flat gradient buffer
        ex = GradientExchange(dist, world, direct=direct)
        ex.start(g[6:])                                             # bucket 0: the tail (11 floats: 9 exchanged directly + 2 left over)
        ex.start(g[:6])                                             # bucket 1
        ex.finish()
        out[direct] = g.tolist()
    assert GradientExchange(dist, world).direct is False            # gloo: plain all-reduce unless asked
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_exchange_buckets_direct_and_allreduce_world3_gloo():
    """The bucketed exchange: direct reduce-scatter (all-to-all of 1/world chunks + rank-ordered shard
    sum) + all-gather, with the remainder that world does not divide all-reduced, equals the plain all-reduce mean, and
    every rank ends with the same bits."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31600 + (os.getpid() % 2000)
    world = 3
    ps = [ctx.Process(target=_rank_exchange, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    want = ((np.arange(17) + 1) * (1 + 4 + 9) / 3.0).astype(np.float32)
    for r in range(world):
        for direct in (True, False):
            assert np.allclose(got[r][direct], want, rtol=1e-6), (r, direct)
        assert got[r][True] == got[0][True]                         # rank-ordered shard sums: identical on every rank


# ------------------------------------------------------------------------------------------------
# the epoch harness on CPU, driven by a test double whose step is the training oracle
# ------------------------------------------------------------------------------------------------
class OracleTrainer:
    """Test double with DanTrainer's surface (backward / apply / state_dict / set_lr), backed by the CPU training oracle --
    test infrastructure only; the product trainer refuses to exist without the HIP extension + GPU."""

    def __init__(self, cfg, hyper, sd):
        from oracle import dan_train_oracle as T
        self.T, self.config, self.hyper = T, cfg, hyper
        self.sd, self.adam, self.step, self._pending = dict(sd), None, 0, None

    def backward(self, planes, targets, dropout_masks=None, seed=0):
        T = self.T
        hp = T.TrainHyper(**{k: getattr(self.hyper, k) for k in T.TrainHyper.__dataclass_fields__})
        B = len(planes[0])
        widths = (self.config.feature_width,) + tuple(self.config.fc_sizes)
        masks = [np.ones((B, w), np.uint8) for w in widths]
        self._pending = T.train_step_oracle(self.sd, self.config, planes, targets, hp, dropout_masks=masks, adam_state=self.adam,
                                            step=self.step + 1)
        r = self._pending
        return {**{k: float(r[k]) for k in ("loss", "bin", "vt", "af", "cov", "vb", "vr")}, "vt_close": r["vt_close"], "bin_close": r["bin_close"]}

    def apply(self):
        r = self._pending
        self.sd = {**self.sd, **{k[4:]: v for k, v in r.items() if k.startswith("new:")}}
        self.adam = {k: v for k, v in r.items() if k.startswith(("m:", "v:"))}
        self.step += 1
        return float(r["grad_norm"])


def test_epoch_harness_and_eval_losses_on_cpu(tmp_path):
    import torch
    from dl4vc_amd import hdf5io
    from dl4vc_amd.config import DanConfig
    from dl4vc_amd.trainer import train_epoch, evaluate, eval_losses, read_indices, split_batch
    from oracle.dan_oracle import random_state_dict, dan_forward_oracle
    from oracle import dan_train_oracle as T
    cfg = DanConfig(reads=8, c_init=8, c_final=8, bottleneck=2, fc_sizes=(8, 4))
    hp = TrainHyper(dropout=0.0)
    recs = _records(24, 8, 77)
    path = str(tmp_path / "train.hdf")
    hdf5io.write_candidates(path, recs)
    sd = random_state_dict(cfg, seed=5, dropout_keys=False)
    assert split_batch(10, 0, 4) == (0, 3) and split_batch(10, 3, 4) == (9, 10) and split_batch(2, 3, 4) == (2, 2)
    with hdf5io.CandidateFile(path) as src:
        got = read_indices(src, np.array([7, 3, 4, 20, 3]))
        assert [bytes(r["vcfrec"]) for r in got] == [bytes(recs[i]["vcfrec"]) for i in (7, 3, 4, 20, 3)]
        tr = OracleTrainer(cfg, hp, sd)
        sm = EasyExampleSampler(len(src), close_keep=0.15, rng=np.random.RandomState(3))
        lines = []
        mean = train_epoch(tr, src, sm, hp, batch_size=10, epoch=1, reads_seed=0, log=lines.append)
        assert tr.step == 3 and np.isfinite(mean["loss"]) and len(lines) == 4 and "Loss:" in lines[0] and "close matches" in lines[-1]
        # max_train_batches N runs batches 0..N (trainer.py:113-115)
        tr2 = OracleTrainer(cfg, hp, sd)
        train_epoch(tr2, src, EasyExampleSampler(len(src), rng=np.random.RandomState(3)), hp, 5, 1, max_batches=1, log=None)
        assert tr2.step == 2

        class Net:                                                     # eval-mode forward = the inference oracle
            config = cfg

            def forward_u8(self, *planes, aux=False):
                return dan_forward_oracle(tr.sd, cfg, *planes)

        out = []
        loss = evaluate(Net(), src, hp, batch_size=7, write=out.append)
        assert np.isfinite(loss) and sum(t.count("\n") for t in out) == 24 and "BP=" in out[0]
    # the host restatement of the loss mix against the oracle's torch version
    b = assemble_training_batch(recs[:9], np.arange(9), 8, non_snp_train_weight=hp.non_snp_train_weight)
    o = dan_forward_oracle(sd, cfg, *b.planes())
    want = T.losses({k: torch.from_numpy(np.asarray(o[k])) for k in ("bin_logits", "vt_logits", "af", "cov", "vb", "vr")}, b.targets,
                    T.TrainHyper(**{k: getattr(hp, k) for k in T.TrainHyper.__dataclass_fields__}))
    mine = eval_losses(o, b.targets, hp)
    for k in ("loss", "bin", "vt", "af", "cov", "vb", "vr"):
        assert abs(mine[k] - float(want[k])) < 1e-5 * max(1.0, abs(float(want[k]))), k


def test_main_refuses_unsupported_training_options():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), ".."))
    import main as cli
    base = ["--train_file", "t.hdf", "--test_file", "x.hdf", "--model_pool_combine_dimension", "0"]
    with pytest.raises(SystemExit, match="augment-single-reads"):
        cli.main(base + ["--augment-single-reads"])
    with pytest.raises(SystemExit, match="rm_var_reads_rate"):
        cli.main(base + ["--rm_var_reads_rate", "0.1"])


def test_batch_prefetcher_workers_yield_the_same_batches_in_order(tmp_path):
    """The loader workers (``--num-data-workers``; main.py:59-60) are spawned processes with their own HDF5 handles; what they
    yield is what the in-process path assembles, in the order asked, for shuffled and ragged index lists."""
    from dl4vc_amd import hdf5io
    from dl4vc_amd.train_data import BatchPrefetcher, read_indices
    from dl4vc_amd.synth import make_labelled_records as make_records
    recs = make_records(24, 20, 900)
    path = str(tmp_path / "train.hdf")
    hdf5io.write_candidates(path, recs)
    order = np.random.RandomState(3).permutation(24)
    lists = [order[0:7], order[7:14], order[14:21], order[21:24]]
    kw = dict(max_reads=12, seed=5, non_snp_train_weight=0.5)
    with BatchPrefetcher(path, workers=0) as p0:
        want = list(p0.batches(iter(lists), **kw))
    with BatchPrefetcher(path, workers=2, depth=2) as p2:
        got = list(p2.batches(iter(lists), **kw))
        again = list(p2.batches(iter(lists[:2]), **kw))             # the pool serves a second epoch
    assert len(got) == len(want) == 4 and len(again) == 2
    for g, w, idx in zip(got, want, lists):
        assert np.array_equal(g.index, idx) and np.array_equal(w.index, idx)
        for a, b in zip(g.planes(), w.planes()):
            assert np.array_equal(a, b)
        for k in w.targets:
            assert np.array_equal(g.targets[k], w.targets[k]), k
        assert g.sites.vcfrec == w.sites.vcfrec and np.array_equal(g.blacklist, w.blacklist)
    with hdf5io.CandidateFile(path) as src:
        r = read_indices(src, np.array([5, 3, 4, 20]))
        assert [bytes(x["name"]) for x in r] == [bytes(recs[i]["name"]) for i in (5, 3, 4, 20)]
