"""Training harness: the epoch loop around the HIP training step.

Stands where ``dl4vc/trainer.py::train`` (trainer.py:64-472) and the epoch loop of ``main.py`` (main.py:151-199) stand:
iterate the sampler's order in batches, assemble planes + targets, one ``DanTrainer.train_step`` per batch, write the
close-example / blacklist flags back into the sampler (trainer.py:258-267), print the reference's progress line
(trainer.py:443-448); after an epoch decay the learning rate (main.py:166), evaluate on the test file (the same loss mix in
eval mode, trainer.py:575-604, and the scored VCF, trainer.py:678-681) and save ``<name>_epoch<N><ext>`` / ``<name>_best<ext>``
checkpoints (utils.py:180-186) holding ``{'epoch','state_dict','best_loss','optimizer'}`` with ``module.``-prefixed keys as
the reference's DataParallel wrapper produces (main.py:194-199).

Data parallelism: one process per GPU; the ``--batch-size`` sites of a step are split over the ranks the way
``nn.DataParallel`` scatters them (main.py:117), every rank runs forward + backward on its share (BatchNorm statistics per
replica, as DataParallel does), the flat gradient buffers are averaged with ONE all-reduce (RCCL over xGMI) and every rank
applies the same Adam update.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Dict, Optional

import numpy as np

from .hdf5io import CandidateFile
from .train_data import assemble_training_batch, read_indices, BatchPrefetcher, EasyExampleSampler   # noqa: F401
from .train import TrainHyper, average_gradients, GradientExchange, base_class_weight_sums

COVERAGE_SCALE_FACTOR = 1.0 / 100.0                       # trainer.py:61
BASE_CLASS_WEIGHT = np.array([0.001, 1., 1., 1., 1., 1., 0.001, 0.001, 1., 0.001])     # trainer.py:312-313


# ------------------------------------------------------------------------------------------------
# the loss mix on host arrays (evaluation only: trainer.py:575-604 runs the same criteria under no_grad)
# ------------------------------------------------------------------------------------------------
def _softmax(x):
    e = np.exp(x - x.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def _focal(logits, target, weight, pos_weight, hp: TrainHyper):
    """objectives.py:77-112 (logits=True), float64."""
    x = np.asarray(logits, np.float64)
    n = x.shape[1]
    y = np.full_like(x, hp.label_smoothing / (n - 1))
    y[np.arange(len(x)), target] = 1.0 - hp.label_smoothing
    ce = weight[:, None] * (np.maximum(x, 0) - x * y + np.log1p(np.exp(-np.abs(x))))
    p = np.clip(_softmax(x), 0.0, 1.0)
    pt = y * p + (1 - y) * (1 - p)
    w = (1 - pt) ** hp.focal_gamma * (pos_weight / pos_weight.sum())
    return float((hp.focal_alpha * w * ce).sum(axis=1).mean())


def eval_losses(out: Dict[str, np.ndarray], targets: Dict[str, np.ndarray], hp: TrainHyper) -> Dict[str, float]:
    """``out``: bin_logits, vt_logits, af, cov, vb, vr of an eval-mode forward (``DanNet.forward_u8(aux=True)``)."""
    w = np.asarray(targets["weight"], np.float64)
    t_bin = (np.asarray(targets["label"]) <= 1).astype(np.int64)                        # trainer.py:134
    t_vt = np.asarray(targets["var_type"]).astype(np.int64)
    bin_loss = _focal(out["bin_logits"], t_bin, w, np.array([hp.fp_train_weight, 1.0]), hp)
    vt_loss = _focal(out["vt_logits"], t_vt, w, np.array([hp.fp_train_weight, 1.0, 1.0]), hp)
    af = np.asarray(out["af"], np.float64).reshape(-1)
    t_af = np.asarray(targets["allele_freq"], np.float64)
    af_loss = float((-w * (t_af * np.maximum(np.log(af), -100) + (1 - t_af) * np.maximum(np.log1p(-af), -100))).mean())
    cov = np.asarray(out["cov"], np.float64).reshape(-1)
    cov_loss = float(((cov - np.asarray(targets["coverage"], np.float64) * COVERAGE_SCALE_FACTOR) ** 2).mean())

    def ce(logits, y):
        x = np.asarray(logits, np.float64)
        lse = np.log(np.exp(x - x.max(axis=1, keepdims=True)).sum(axis=1)) + x.max(axis=1)
        wy = BASE_CLASS_WEIGHT[y]
        return float((wy * (lse - x[np.arange(len(x)), y])).sum() / wy.sum())

    vb_loss = ce(out["vb"], np.asarray(targets["var_base_enum"]).astype(np.int64))
    vr_loss = ce(out["vr"], np.asarray(targets["var_ref_enum"]).astype(np.int64))
    loss = bin_loss * hp.binary_weight + (vt_loss + af_loss * hp.aux_allele_weight + cov_loss +
                                          (vb_loss + vr_loss) * hp.aux_bases_weight) * hp.aux_weight   # trainer.py:426-427
    return {"loss": loss, "bin": bin_loss, "vt": vt_loss, "af": af_loss, "cov": cov_loss, "vb": vb_loss, "vr": vr_loss}


# ------------------------------------------------------------------------------------------------
def split_batch(n: int, rank: int, world: int):
    """The slice of a batch of ``n`` sites that ``nn.DataParallel``'s scatter (torch.chunk along dim 0) gives replica ``rank``:
    chunks of ceil(n / world) sites, so a short last batch leaves the LAST replicas without sites (49 sites over 8 replicas:
    seven chunks of 7) -- DataParallel then simply runs fewer replicas; here such a rank sits the step out (``train_epoch``)."""
    chunk = -(-n // world)
    lo = min(n, rank * chunk)
    return lo, min(n, lo + chunk)


def train_epoch(trainer, source: CandidateFile, sampler: EasyExampleSampler, hyper: TrainHyper, batch_size: int, epoch: int,
                reads_seed: int = 0, max_batches: int = 0, keep_candidate_af: bool = True, log: Optional[Callable] = print,
                rank: int = 0, world: int = 1, all_reduce=None, gather=None, log_interval: int = 1,
                exchange: Optional[GradientExchange] = None, prefetcher: Optional[BatchPrefetcher] = None) -> Dict[str, float]:
    """One pass of ``trainer.train`` (trainer.py:64-472).  ``trainer``: anything with ``backward`` / ``apply`` (and
    ``grad_tensor`` when ``world > 1``) -- ``DanTrainer`` on the GPU.  With ``exchange`` (and a trainer that has the
    ``backward_begin`` / ``wait_bucket`` / ``backward_end`` split) the FC-side gradient bucket is averaged while the
    convolution layers' backward still runs; otherwise one ``all_reduce`` of the flat buffer after the backward pass.
    ``prefetcher``: loader workers assembling this rank's batches ahead of the GPU (``--num-data-workers``); without it the
    batches are assembled in this process between steps.  Returns the epoch's mean losses."""
    order = sampler.epoch()
    cfg = trainer.config
    tot = {k: 0.0 for k in ("loss", "bin", "vt", "af", "cov", "vb", "vr")}
    n_batches = close_n = items = 0
    t0 = time.perf_counter()
    plan = []                                                          # (batch number, offset in the order, this rank's indices)
    for b, lo in enumerate(range(0, len(order), batch_size)):
        if max_batches > 0 and b > max_batches:                       # trainer.py:113-115 (same off-by-one)
            break
        idx = order[lo:lo + batch_size]
        a, e = split_batch(len(idx), rank, world)
        plan.append((b, lo, idx[a:e]))                                 # possibly empty: the rank sits that step out (below)
    # the read subset of a pileup deeper than max_reads is redrawn every epoch, as the reference's is (dataset.py:274-281
    # draws from numpy's global generator, which the epoch's shuffles advance); evaluation and inference keep reads_seed
    draw_seed = reads_seed + epoch * len(source)
    # dropout: DataParallel's replicas draw independent masks (each its own device generator); fold the rank in
    drop_seed = (reads_seed + epoch) * world + rank
    kwargs = dict(max_reads=cfg.reads, seed=draw_seed, non_snp_train_weight=hyper.non_snp_train_weight,
                  keep_candidate_af=keep_candidate_af, use_q=cfg.use_q, use_strand=cfg.use_strand)
    busy = [m for _, _, m in plan if len(m)]
    if prefetcher is not None:
        made = prefetcher.batches(iter(busy), **kwargs)
    else:
        made = (assemble_training_batch(read_indices(source, m), m, **kwargs) for m in busy)
    stream = (next(made) if len(m) else None for _, _, m in plan)
    def global_normalisers(batch):
        """Full-batch loss normalisers / ranks (one tiny all-reduce): with them the average of the ranks' gradients is the
        full-batch gradient nn.DataParallel computes, also for unequal shards (include/dl4vc_dan_train.h)."""
        import torch
        dist = exchange.dist
        sums = base_class_weight_sums(batch.targets) if batch is not None else np.zeros(3)
        t = torch.tensor(sums, dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t)
        g = t.cpu().numpy() / world
        if batch is not None:
            trainer.set_global_batch(g[0], g[1], g[2])

    def sit_out():
        """A step in which this rank has no sites (the short last batch of an epoch): DataParallel runs fewer replicas; here
        the rank contributes a ZERO gradient to the average, takes part in every collective of the step and applies the same
        update as the others, so the replicas stay identical and nobody waits on a rank that left."""
        grad = trainer.grad_tensor()
        grad.zero_()
        if grad.is_cuda:
            import torch
            torch.cuda.synchronize()                                   # the exchange runs on a side stream
        if exchange is not None:
            (o0, n0), (o1, n1) = trainer.grad_buckets()
            exchange.start(grad[o0:o0 + n0])
            exchange.start(grad[o1:o1 + n1])
            exchange.finish()
        else:
            average_gradients(grad, world, all_reduce)
        out = {k: 0.0 for k in tot}
        out["vt_close"] = np.zeros(0, bool)
        return out

    split = hasattr(trainer, "backward_begin")
    feed = zip(plan, stream)
    nxt = next(feed, None)
    while nxt is not None:
        (b, lo, mine), batch = nxt
        if world > 1 and exchange is not None and hasattr(trainer, "set_global_batch"):
            global_normalisers(batch)
        if batch is None:
            if world <= 1:
                raise RuntimeError("empty training batch")
            out = sit_out()
            nxt = next(feed, None)
        elif split:
            # enqueue the step, then take delivery of the next batch (worker hand-over, unpickling) while the device works
            trainer.backward_begin(batch.planes(), batch.targets, seed=drop_seed)
            nxt = next(feed, None)
            if world > 1 and exchange is not None:
                grad = trainer.grad_tensor()
                (o0, n0), (o1, n1) = trainer.grad_buckets()
                trainer.wait_bucket(0)
                exchange.start(grad[o0:o0 + n0])
                out = trainer.backward_end()
                exchange.start(grad[o1:o1 + n1])
                exchange.finish()
            else:
                out = trainer.backward_end()
                if world > 1:
                    average_gradients(trainer.grad_tensor(), world, all_reduce)
        else:
            out = trainer.backward(batch.planes(), batch.targets, seed=drop_seed)
            nxt = next(feed, None)
            if world > 1:
                average_gradients(trainer.grad_tensor(), world, all_reduce)
        trainer.apply()
        flags = [(mine, out["vt_close"], batch.blacklist if batch is not None else np.zeros(0, bool))]
        if world > 1 and gather is not None:
            flags = gather(flags[0])
        for ids, close, black in flags:                                # trainer.py:263-267
            sampler.update_close(ids, close)
            sampler.update_blacklist(ids, black)
            close_n += int(np.sum(close))
            items += len(ids)
        for k in tot:
            tot[k] += out[k]
        n_batches += 1
        if log and rank == 0 and b % max(log_interval, 1) == 0:
            n = n_batches
            log("  Elapsed ({:.02e}s) [{}/{} ({:.0f}%)]  Loss: {:.6f}  Total: {:.6f}| bin: {:.5f} vt: {:.5f} dlt: {:.5f} af: {:.5f} cov: {:.5f} bases: {:.5f}".format(
                time.perf_counter() - t0, b * batch_size, len(order), 100.0 * lo / max(len(order), 1), out["loss"], tot["loss"] / n,
                tot["bin"] / n, tot["vt"] / n, 0.0, tot["af"] / n, tot["cov"] / n, (tot["vb"] + tot["vr"]) / n))
            t0 = time.perf_counter()
    if log and rank == 0:
        log('%d/%d [%.2f%%] "close matches" within %.2f * %.5f (label smoothing) of true label' %
            (close_n, items, close_n / max(items, 1) * 100.0, hyper.close_match_window, hyper.label_smoothing))
    return {k: v / max(n_batches, 1) for k, v in tot.items()}


def evaluate(net, source: CandidateFile, hyper: TrainHyper, batch_size: int, write: Optional[Callable[[str], None]] = None,
             reads_seed: int = 0, max_batches: int = 0, indices: Optional[np.ndarray] = None,
             prefetcher: Optional[BatchPrefetcher] = None, rank: int = 0, world: int = 1, reduce: bool = True):
    """Eval-mode pass over the test file (trainer.py:509-681): mean over batches of the loss mix; optionally the scored VCF
    records.  ``net``: a ``DanNet`` (running BatchNorm statistics, no dropout).  ``world > 1``: this rank takes the
    ``rank``-th contiguous run of the batches (the runs concatenated in rank order are the single-process order);
    ``reduce=False`` returns (sum of batch losses, batches) for the caller to sum over the ranks."""
    from .vcf import scored_record
    cfg = net.config
    idx_all = np.arange(len(source)) if indices is None else np.asarray(indices)
    total, n_batches = 0.0, 0
    plan = []
    for b, lo in enumerate(range(0, len(idx_all), batch_size)):
        if max_batches > 0 and b > max_batches:                       # trainer.py:513-515
            break
        plan.append(idx_all[lo:lo + batch_size])
    if world > 1:
        plan = plan[len(plan) * rank // world:len(plan) * (rank + 1) // world]
    kwargs = dict(max_reads=cfg.reads, seed=reads_seed, non_snp_train_weight=hyper.non_snp_train_weight, use_q=cfg.use_q,
                  use_strand=cfg.use_strand)
    if prefetcher is not None:
        stream = prefetcher.batches(iter(plan), **kwargs)
    else:
        stream = (assemble_training_batch(read_indices(source, idx), idx, **kwargs) for idx in plan)
    for batch in stream:
        out = net.forward_u8(*batch.planes(), aux=True)
        total += eval_losses(out, batch.targets, hyper)["loss"]
        n_batches += 1
        if write:
            write("".join(scored_record(r, bp, v) + "\n" for r, bp, v in zip(batch.sites.vcfrec, out["bp"], out["vt_prob"])))
    if not reduce:
        return total, n_batches
    return total / max(n_batches, 1)


def save_checkpoint(state: dict, is_best: bool, filename: str = "checkpoint.pth.tar") -> None:
    """utils.py:180-186."""
    import torch
    base, ext = os.path.splitext(filename)
    torch.save(state, "{}_epoch{}{}".format(base, state["epoch"], ext))
    if is_best:
        torch.save(state, "{}_best{}".format(base, ext))


def reference_parameter_order(cfg, fc_keys=("conv2hidden.1", "conv2hidden.4")):
    """``[name for name, _ in Basic2DNet.named_parameters()]`` -- the index space of the reference's ``optim.Adam(model.
    parameters())`` (main.py:116): the module's own two parameters first (model.py:429-431), then the children in the order
    the constructor registers them (model.py:143,265-271,377,406-415).  Pinned against the reference's own list in
    tests/golden/train_*.npz (tests/test_train_data.py)."""
    names = ["bin_output_weights", "vt_output_weights", "embeddings.weight"]
    n = cfg.layers
    names += ["conv1D_layers.%d.%s" % (i, p) for i in range(n) for p in ("weight", "bias")]
    names += ["bn1D_layers.%d.%s" % (i, p) for i in range(n) for p in ("weight", "bias")]
    if cfg.bottleneck > 0:
        names += ["conv1D_bottleneck_layers.%d.%s" % (i, p) for i in range(n) for p in ("weight", "bias")]
        names += ["conv1D_compression_layers.%d.%s" % (i, p) for i in range(n) for p in ("weight", "bias")]
    n_res = sum(1 for l in range(1, n + 1) if cfg.is_residual(l))                       # model.py:246-252
    names += ["residual_conv_layers.%d.%s" % (i, p) for i in range(n_res) for p in ("weight", "bias")]
    names += ["%s.%s" % (k, p) for k in fc_keys for p in ("weight", "bias")]
    for head in ("fcHidden2BinTarget", "fcHidden2VT", "fcHidden2AF", "fcHidden2Coverage", "fcHidden2VB", "fcHidden2VR"):
        names += [head + ".weight", head + ".bias"]
    return names


def optimizer_state(trainer) -> dict:
    """``optimizer.state_dict()`` of the reference's Adam (main.py:116,198): ``state`` by parameter index (step, exp_avg,
    exp_avg_sq) for every parameter a gradient reaches, and one ``param_groups`` entry -- what
    ``torch.optim.Adam.load_state_dict`` accepts."""
    import torch
    hp = trainer.hyper
    fc_keys = getattr(trainer, "_fc_keys", ("conv2hidden.1", "conv2hidden.4"))
    names = reference_parameter_order(trainer.config, fc_keys)
    step = int(trainer.query("step"))
    state = {}
    for i, name in enumerate(names):
        ours = name
        for j, k in enumerate(fc_keys):
            if name.startswith(k + "."):
                ours = "fc.%d.%s" % (j, name.rsplit(".", 1)[1])
        if step == 0:
            break
        try:
            m, v = trainer.tensor("m:" + ours), trainer.tensor("v:" + ours)
        except (KeyError, RuntimeError):
            continue                                              # no gradient reaches it (mixing scalars; BN affine with BN off)
        state[i] = {"step": step, "exp_avg": torch.from_numpy(np.ascontiguousarray(m)),
                    "exp_avg_sq": torch.from_numpy(np.ascontiguousarray(v))}
    group = {"lr": float(hp.lr), "betas": (float(hp.beta1), float(hp.beta2)), "eps": float(hp.adam_eps), "weight_decay": 0,
             "amsgrad": False, "params": list(range(len(names)))}
    return {"state": state, "param_groups": [group], "param_names": names}


def checkpoint_state(trainer, epoch: int, best_loss: float) -> dict:
    """The dict the reference saves (main.py:194-199): DataParallel-prefixed state_dict + the Adam state in torch's own
    ``optimizer.state_dict()`` shape.  (As in the reference, ``--modelload`` restores the parameters only -- main.py:121-124
    never calls ``optimizer.load_state_dict`` -- so a resumed run restarts Adam's moments from zero.)"""
    import torch
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in trainer.state_dict(prefix="module.").items()}
    return {"epoch": epoch, "state_dict": sd, "best_loss": best_loss, "optimizer": optimizer_state(trainer)}
