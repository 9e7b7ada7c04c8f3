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


@pytest.mark.parametrize("world", [3, 8])
def test_gradient_exchange_buckets_direct_and_allreduce_gloo(world):
    """The bucketed exchange: direct reduce-scatter (all-to-all of 1/world chunks + rank-ordered shard
    sum) + all-gather, with the remainder that world does not divide all-reduced, equals the plain all-reduce mean, and
    every rank ends with the same bits.  World 3 and world 8 (BASELINE config 4's node: bucket 0 = 11 floats = 8 exchanged
    directly + 3 left over, bucket 1 = 6 floats < world: all of it through the remainder path)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31600 + (os.getpid() % 2000) + world
    ps = [ctx.Process(target=_rank_exchange, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    want = ((np.arange(17) + 1) * sum((r + 1) ** 2 for r in range(world)) / float(world)).astype(np.float32)
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


# ------------------------------------------------------------------------------------------------
# a short last batch leaves the last ranks without sites: nobody hangs, the replicas stay identical
# ------------------------------------------------------------------------------------------------
class ToyTrainer:
    """DanTrainer's data-parallel surface on a five-parameter toy: the per-site gradient is a fixed function of the site's
    planes, a rank's gradient is the sum over its sites divided by the full-batch normaliser it was handed
    (set_global_batch), apply is plain SGD.  Then the average of the ranks' gradients is the full-batch mean gradient
    whatever the split -- what include/dl4vc_dan_train.h promises for the real step."""

    def __init__(self, cfg, hyper):
        import torch
        self.config, self.hyper = cfg, hyper
        self.p = torch.zeros(5, dtype=torch.float64)
        self.g = torch.zeros(5, dtype=torch.float64)
        self.per_rank, self.steps, self.sat_out = None, 0, 0

    def set_global_batch(self, sites_per_rank, vb, vr):
        self.per_rank = float(sites_per_rank)

    def _site_grad(self, planes):
        import torch
        reads, qual = planes[0].astype(np.float64), planes[1].astype(np.float64)
        f = np.stack([reads.mean(axis=(1, 2)), qual.mean(axis=(1, 2)), reads.std(axis=(1, 2)), (reads[:, 0] % 3).mean(axis=1),
                      np.ones(len(reads))], axis=1)
        return torch.from_numpy(f.sum(axis=0))

    def backward_begin(self, planes, targets, dropout_masks=None, seed=0):
        n = len(planes[0])
        self.g.copy_(self._site_grad(planes) / (self.per_rank if self.per_rank else n))
        self._n = n

    def wait_bucket(self, b):
        pass

    def backward_end(self):
        return {**{k: 1.0 for k in ("loss", "bin", "vt", "af", "cov", "vb", "vr")}, "vt_close": np.zeros(self._n, bool)}

    def grad_tensor(self):
        return self.g

    def grad_buckets(self):
        return [(2, 3), (0, 2)]

    def apply(self):
        self.p -= 0.1 * self.g
        self.steps += 1
        return float(self.g.norm())


def _rank_short_tail(rank, world, port, path, q):
    import torch.distributed as dist
    from dl4vc_amd import hdf5io
    from dl4vc_amd.config import DanConfig
    from dl4vc_amd.trainer import train_epoch
    from dl4vc_amd.train import GradientExchange
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = DanConfig(reads=8, c_init=8, c_final=8, bottleneck=2, fc_sizes=(8, 4))
    hp = TrainHyper(dropout=0.0)
    tr = ToyTrainer(cfg, hp)
    flags = []

    def gather(item):
        out = [None] * world
        dist.all_gather_object(out, item)
        return out

    with hdf5io.CandidateFile(path) as src:
        sm = EasyExampleSampler(len(src), close_keep=1.0, rng=np.random.RandomState(3))
        orig = sm.update_close
        sm.update_close = lambda ids, close: (flags.append(list(map(int, ids))), orig(ids, close))[1]
        for epoch in (1, 2):
            train_epoch(tr, src, sm, hp, batch_size=4, epoch=epoch, log=None, rank=rank, world=world,
                        all_reduce=dist.all_reduce if world > 1 else None, gather=gather if world > 1 else None,
                        exchange=GradientExchange(dist, world) if world > 1 else None)
    q.put((rank, tr.p.tolist(), tr.steps, sorted(i for f in flags for i in f)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("ranks", [2, 8])
def test_short_last_batch_leaves_ranks_without_sites_gloo(tmp_path, ranks):
    """9 sites in batches of 4 over 2 ranks: the last batch holds ONE site, so rank 1 has nothing in it (DataParallel's
    torch.chunk split, main.py:117: it then runs one replica).  Both ranks finish both epochs (no collective left waiting),
    take the same number of optimiser steps, end with bit-identical parameters, and those equal the single-process run --
    the averaged gradient is the full-batch gradient for the 2+2, 2+2 and 1+0 splits alike.  At 8 ranks (the node of
    BASELINE config 4) a batch of 4 leaves ranks 4-7 out of EVERY step and the one-site batch ranks 1-7: they sit the steps
    out, join every exchange and stay bit-identical."""
    import torch.multiprocessing as mp
    from dl4vc_amd import hdf5io
    path = str(tmp_path / "train.hdf")
    hdf5io.write_candidates(path, _records(9, 8, 77))
    ctx = mp.get_context("spawn")
    res = {}
    for world in (1, ranks):
        q = ctx.Queue()
        port = 33700 + (os.getpid() % 2000) + world
        ps = [ctx.Process(target=_rank_short_tail, args=(r, world, port, path, q)) for r in range(world)]
        for p in ps:
            p.start()
        got = [q.get(timeout=180) for _ in ps]
        for p in ps:
            p.join(60)
            assert p.exitcode == 0
        res[world] = {r: (p_, s, f) for r, p_, s, f in got}
    p1, s1, f1 = res[1][0]
    assert s1 == 6                                                   # 3 batches x 2 epochs
    for r in range(ranks):
        p2, s2, f2 = res[ranks][r]
        assert s2 == 6 and f2 == f1 == sorted(list(range(9)) * 2)    # every site's flags reached every rank's sampler, once per epoch
        assert np.allclose(p2, p1, rtol=1e-12, atol=1e-12), (r, p2, p1)
        assert res[ranks][r][0] == res[ranks][0][0]                  # replicas identical bit for bit


def test_reference_parameter_order_and_optimizer_state_shape():
    """The index space of the reference's Adam state (optim.Adam(model.parameters()), main.py:116,198) restated from the
    configuration equals the reference's own named_parameters() order -- the order of the parameter keys of the state dict
    its training loop left in tests/golden/train_*.npz -- for all six structures; the checkpoint's 'optimizer' entry loads
    into a torch Adam over tensors of those shapes."""
    import torch
    from golden_util import load_train_case, train_cases
    from dl4vc_amd.config import DanConfig
    from dl4vc_amd.trainer import reference_parameter_order, optimizer_state
    buffers = ("pe", "running_mean", "running_var", "num_batches_tracked")
    for case in train_cases():
        spec, hyper, w, steps, final, adam, close = load_train_case(case)
        keys = DanConfig.__dataclass_fields__.keys()
        cfg = DanConfig(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in spec.items() if k in keys})
        want = [k for k in final if not k.endswith(buffers)]
        fc = sorted({k.rsplit(".", 1)[0] for k in want if k.startswith("conv2hidden.")}, key=lambda k: int(k.split(".")[1]))
        assert reference_parameter_order(cfg, tuple(fc)) == want, case

    class Stub:                                                      # DanTrainer's read-only surface, moments = the fixture's
        def __init__(self, cfg, hp, adam, fc):
            self.config, self.hyper, self._fc_keys, self.adam = cfg, hp, fc, adam

        def query(self, what):
            return 2

        def tensor(self, name):
            kind, base = name.split(":", 1)
            if base.startswith("fc."):
                base = "%s.%s" % (self._fc_keys[int(base.split(".")[1])], base.split(".")[2])
            return self.adam["adam_%s:%s" % (kind, base)]            # KeyError where no gradient reaches

    spec, hyper, w, steps, final, adam, close = load_train_case("train_small")
    cfg = DanConfig(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in spec.items() if k in DanConfig.__dataclass_fields__})
    hp = TrainHyper(**{k: v for k, v in hyper.items() if k in TrainHyper.__dataclass_fields__})
    fc = tuple(sorted({k.rsplit(".", 1)[0] for k in final if k.startswith("conv2hidden.")}, key=lambda k: int(k.split(".")[1])))
    od = optimizer_state(Stub(cfg, hp, adam, fc))
    names = od.pop("param_names")
    params = [torch.nn.Parameter(torch.from_numpy(np.array(final[k], dtype=np.float32))) for k in names]
    opt = torch.optim.Adam(params, lr=1.0)
    opt.load_state_dict(od)                                          # what a tool written against the reference's checkpoints does
    st = opt.state_dict()["state"]
    i = names.index("conv1D_layers.2.weight")
    assert np.array_equal(st[i]["exp_avg"].numpy(), adam["adam_m:conv1D_layers.2.weight"]) and float(st[i]["step"]) == 2
    assert names.index("bin_output_weights") not in st and opt.param_groups[0]["lr"] == hp.lr
