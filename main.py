#!/usr/bin/env python3
"""Drop-in for the reference's ``main.py`` (main.py:47-229): inference-only mode (branch :213-222) and, with
``--train_file``, training + per-epoch evaluation (branch :151-199).

    python main.py <the flags call_variants.sh passes> --test_file X.hdf --modelload CKPT \
        --save_vcf_records --save_vcf_records_file OUT/model_test.vcf --sample_vcf OUT/candidates.vcf

Reads the candidate HDF5 (schema of tools/convert_bam_single_reads.py), scores every site with the
MI355X-native DAN forward and writes ``OUT/epoch1_model_test.vcf`` exactly where and how the reference does
(dl4vc/utils.py:146-178).  ``--gpus N`` starts one process per GPU over contiguous site shards and
concatenates the part files on the host; there is no collective on this path.  With ``--train_file`` the flags of
train_variant_caller.sh:101-151 drive ``dl4vc_amd/trainer.py`` (one process per GPU, one RCCL all-reduce of the flat
gradient buffer per step); training options outside the published script's path are refused, never ignored.
"""
from __future__ import annotations

import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from arguments import create_arg_parser                       # noqa: E402


def child_devices(n: int):
    """HIP_VISIBLE_DEVICES value of each of the ``n`` shard processes: the g-th entry of the PARENT's device mask
    (HIP_VISIBLE_DEVICES, else CUDA_VISIBLE_DEVICES, which HIP honours too), or plain g without a mask.  A
    ROCR_VISIBLE_DEVICES mask needs no handling: HIP indices are already relative to it and the children inherit it."""
    if os.environ.get("DL4VC_FORCE_DEVICE0"):                 # rehearse the multi-process path on a one-GPU box (tests)
        return ["0"] * n
    mask = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES"))
    if mask is None:
        return [str(g) for g in range(n)]
    have = [d.strip() for d in mask.split(",") if d.strip()]
    if len(have) < n:
        raise SystemExit("--gpus %d but the device mask '%s' lists only %d device(s)" % (n, mask, len(have)))
    return have[:n]


def wait_children(procs, poll_s: float = 0.2):
    """Return codes of the child processes; as soon as ONE exits non-zero the others are terminated (a rank that died leaves
    its siblings blocked in a collective until the backend's timeout -- and a parent waiting on them in rank order blocked
    with them)."""
    rcs = [None] * len(procs)
    failed = False
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        if not failed and any(rc not in (None, 0) for rc in rcs):
            failed = True
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    p.terminate()
            deadline = time.time() + 20.0
            while time.time() < deadline and any(p.poll() is None for p in procs):
                time.sleep(poll_s)
            for p in procs:
                if p.poll() is None:
                    p.kill()
        if any(rc is None for rc in rcs):
            time.sleep(poll_s)
    return rcs


def train_main(args, argv) -> int:
    """main.py:60-80,114-117,151-199: train for --epochs, evaluate on --test_file after every epoch, save checkpoints."""
    import numpy as np
    assert args.train_file[-3:] == "hdf", "Train dataset must be in HDF format"                   # main.py:66
    assert args.test_file[-3:] == "hdf", "Test dataset must be in HDF format"
    refused = [("--augment-single-reads", args.augment_single_reads), ("--augment-reference", args.augment_reference),
               ("--reads-dynamic-downsample-rate", args.reads_dynamic_downsample_rate > 0), ("--rm_var_reads_rate", args.rm_var_reads_rate > 0),
               ("--rm_non_var_reads_rate", args.rm_non_var_reads_rate > 0),
               ("--training_use_directional_augmentation", args.training_use_directional_augmentation),
               ("--train-trust-region-table", bool(args.train_trust_region_table)), ("--gatk-table", bool(getattr(args, "gatk_table", "")))]
    bad = [n for n, on in refused if on]
    if bad:
        raise SystemExit("training option(s) %s are not supported (off in train_variant_caller.sh:101-151): refusing rather than "
                         "silently training something else" % ", ".join(bad))
    if args.precision != "fp32":
        raise SystemExit("training runs in fp32 only")
    from dl4vc_amd.config import DanConfig
    from dl4vc_amd.train import DanTrainer, TrainHyper
    from dl4vc_amd.trainer import train_epoch, evaluate, save_checkpoint, checkpoint_state
    from dl4vc_amd.train_data import EasyExampleSampler
    from dl4vc_amd.hdf5io import CandidateFile
    from dl4vc_amd.model import DanNet, load_checkpoint
    from dl4vc_amd.vcf import start_scored_vcf, scored_vcf_path
    from dl4vc_amd.inference import select_sites
    from dl4vc_amd.shard import check_replicas_agree
    from dl4vc_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus > 1 and world == 1:
        # one process per GPU (the reference: one process, nn.DataParallel over args.gpus devices, main.py:117)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        devices = child_devices(args.gpus)
        procs = []
        for g in range(args.gpus):
            env = dict(os.environ, HIP_VISIBLE_DEVICES=devices[g], HSA_ENABLE_IPC_MODE_LEGACY="0", RANK=str(g), WORLD_SIZE=str(args.gpus),
                       LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.pop("CUDA_VISIBLE_DEVICES", None)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
        rcs = wait_children(procs)
        if any(rcs):
            raise SystemExit("training rank failed: %s" % rcs)
        return 0
    dist = all_reduce = gather = exchange = None
    if world > 1:
        import torch
        import torch.distributed as dist
        backend = os.environ.get("DL4VC_DIST_BACKEND", "nccl")        # (tests rehearse two ranks on one GPU over gloo)
        torch.cuda.set_device(0)
        # second line of defence behind the sharded evaluation below: a rank may legitimately wait long for another (check-
        # point writes, uneven loader start-up), and the backend's default collective timeout (10 min on RCCL) would kill the run
        import datetime
        dist.init_process_group(backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=float(os.environ.get("DL4VC_DIST_TIMEOUT_S", "7200"))))
        all_reduce = dist.all_reduce
        from dl4vc_amd.train import GradientExchange
        exchange = GradientExchange(dist, world)                     # bucketed, overlapped with the backward pass

        def gather(item):
            out = [None] * world
            dist.all_gather_object(out, item)
            return out

    cfg = DanConfig.from_args(args)
    hyper = TrainHyper.from_args(args)
    print("Train on %d rank(s); lr %s, batch %d, dropout %s" % (world, hyper.lr, args.batch_size, hyper.dropout))
    per_rank = -(-args.batch_size // world)
    trainer = DanTrainer(cfg, hyper, max_batch=per_rank, device_id=0)
    if args.modelload:
        print("Loading model checkpoint from {}".format(args.modelload))
        trainer.load_state_dict(load_checkpoint(args.modelload))
    else:
        trainer.load_state_dict(synth.torch_default_init(cfg, seed=args.seed, dropout_keys=hyper.dropout > 0))
    best_loss = None
    from dl4vc_amd.train_data import BatchPrefetcher
    # loader workers (main.py:59-60: DataLoader(num_workers=args.num_data_workers)); 0 = assemble in this process
    with CandidateFile(args.train_file) as train_src, CandidateFile(args.test_file) as test_src, \
            BatchPrefetcher(args.train_file, args.num_data_workers) as train_loader, \
            BatchPrefetcher(args.test_file, args.num_data_workers) as test_loader:
        holdout = None
        if args.train_holdout_chromosomes:
            holdout = np.zeros(len(train_src), bool)
            holdout[select_sites(args.train_file, args.train_holdout_chromosomes)] = True
        # every rank draws the same epoch order (same seed) and takes its DataParallel-style share of every batch
        plain = args.close_examples_sample_rate >= 1.0                # main.py:72-77: then a plain shuffled loader, no sampler
        if plain:
            print("keeping all examples -- no close example down-sampling")
            if holdout is not None and holdout.any():
                print("WARNING: as in the reference (main.py:75-77), without the easy-example sampler the %d held-out sites are NOT "
                      "skipped; use --close_examples_sample_rate < 1 to hold them out" % int(holdout.sum()))
        sampler = EasyExampleSampler(len(train_src), close_keep=min(1.0, args.close_examples_sample_rate), holdout=holdout,
                                     rng=np.random.RandomState(args.seed), plain=plain)
        test_idx = select_sites(args.test_file, args.test_holdout_chromosomes) if args.test_holdout_chromosomes else None
        for epoch in range(1, args.epochs + 1):
            s = time.time()
            print("Train Epoch: %d lr: [%s] on %d GPUs!" % (epoch, trainer.hyper.lr, world))
            train_epoch(trainer, train_src, sampler, hyper, args.batch_size, epoch, reads_seed=args.reads_seed,
                        max_batches=args.max_train_batches, keep_candidate_af=args.aux_keep_candidate_af, rank=rank, world=world,
                        all_reduce=all_reduce, gather=gather, exchange=exchange, prefetcher=train_loader,
                        log_interval=args.log_interval,
                        log=lambda m: print(m, end="\r"))
            print("\n\tTime elapsed for training {:.4f}\n".format(time.time() - s), flush=True)
            s_eval = time.time()
            trainer.set_lr(trainer.hyper.lr * args.lr_decay)                                      # main.py:166
            if epoch <= args.epochs_skip_eval:
                print("Skipping eval for epoch %d" % epoch)
                continue
            # evaluation is SHARDED over the ranks (the reference evaluates under the same DataParallel model, trainer.py:509-681):
            # rank r scores a contiguous run of the test batches with the parameters every rank holds (rank 0's BatchNorm
            # running statistics, broadcast -- DataParallel keeps replica 0's), the loss sums are reduced, the record text is
            # concatenated in rank order = batch order.  No rank idles in a barrier while rank 0 walks a genome-scale file.
            # Only the BatchNorm running statistics can differ between ranks (parameters are identical behind the averaged step;
            # statistics are per replica): those few vectors are broadcast, not the pickled 311-MB state dict.
            state = trainer.state_dict()
            if dist is not None:
                import torch
                on_gpu = dist.get_backend() == "nccl"

                def all_reduce_max(vec):
                    t = torch.from_numpy(np.asarray(vec, np.float64).copy())
                    if on_gpu:
                        t = t.cuda()
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    return t.cpu().numpy()
                # the assumption the broadcast below rests on, checked: one all-reduce of six numbers per evaluation
                check_replicas_agree(state, all_reduce_max, "epoch %d evaluation" % epoch)
                for key in sorted(k for k in state if k.startswith("bn1D_layers.") and
                                  k.rsplit(".", 1)[1] in ("running_mean", "running_var", "num_batches_tracked")):
                    v = np.ascontiguousarray(state[key])
                    t = torch.from_numpy(v.copy())
                    if on_gpu:
                        t = t.cuda()
                    dist.broadcast(t, src=0)
                    state[key] = t.cpu().numpy().astype(v.dtype).reshape(v.shape)
            net = DanNet(cfg, device_id=0, max_batch=args.test_batch_size).load_state_dict(state)
            out_path = None
            if args.save_vcf_records:
                assert args.save_vcf_records_file != "", "Need a valid filename for args.save_vcf_records_file to save records"
                out_path = scored_vcf_path(args.save_vcf_records_file, epoch)
            part = (out_path + ".part%d" % rank) if (out_path and world > 1) else None
            out = None
            if out_path and rank == 0:
                if args.sample_vcf:
                    start_scored_vcf(args.sample_vcf, args.save_vcf_records_file, epoch)
                else:
                    open(out_path, "w").close()
            if out_path:
                out = open(part, "w") if part else open(out_path, "a")
            loss_sum, n_eval = evaluate(net, test_src, hyper, args.test_batch_size, write=out.write if out else None,
                                        reads_seed=args.reads_seed, max_batches=args.max_test_batches, indices=test_idx,
                                        prefetcher=test_loader, rank=rank, world=world, reduce=False)
            if out:
                out.close()
            net.close()
            if dist is not None:
                import torch
                t = torch.tensor([loss_sum, float(n_eval)], dtype=torch.float64,
                                 device="cuda" if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(t)
                loss_sum, n_eval = float(t[0]), int(t[1])
            curloss = loss_sum / max(n_eval, 1)
            if rank == 0:
                if part:
                    with open(out_path, "a") as dst:
                        for r in range(world):
                            with open(out_path + ".part%d" % r) as src:
                                dst.write(src.read())
                            os.remove(out_path + ".part%d" % r)
                print("\nTest set: Average loss: {:.6f}\n".format(curloss))
                is_best = best_loss is None or curloss < best_loss
                best_loss = curloss if best_loss is None else min(curloss, best_loss)
                save_checkpoint(checkpoint_state(trainer, epoch, best_loss), is_best, args.modelsave)   # main.py:194-199
            print("\tTime elapsed for inference/testing {:.4f}".format(time.time() - s_eval))
            print("\tTime elapsed overall {:.4f}\n".format(time.time() - s), flush=True)
            if dist is not None:
                dist.barrier()
    trainer.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0


def main(argv=None) -> int:
    args = create_arg_parser().parse_args(argv)
    print(args)
    if args.train_file:
        return train_main(args, list(argv if argv is not None else sys.argv[1:]))
    assert args.test_file[-3:] == "hdf", "Test dataset must be in HDF format"                     # main.py:84
    print("\n\nRunning in inference only mode...\n\n")
    assert args.modelload is not None, "--modelload argument is required when running in inference only mode"   # main.py:215
    from dl4vc_amd.config import DanConfig
    from dl4vc_amd.shard import parse_shard, part_path, concat_parts
    from dl4vc_amd.vcf import scored_vcf_path, start_scored_vcf

    cfg = DanConfig.from_args(args)                           # rejects unsupported model options loudly
    if args.precision not in ("fp32", "bf16x3", "bf16"):
        raise SystemExit("--precision must be fp32, bf16x3 or bf16")
    if args.conv_algo not in ("auto", "direct", "winograd"):
        raise SystemExit("--conv-algo must be auto, direct or winograd")
    import dataclasses
    cfg = dataclasses.replace(cfg, precision=("fp32", "bf16x3", "bf16").index(args.precision),
                              conv_algo=("auto", "direct", "winograd").index(args.conv_algo),
                              skip_empty_rows=not args.compute_empty_rows)
    # Data-path flags that change WHICH sites reach the VCF: honoured or refused, never silently ignored.
    if args.shuffle_test:
        # main.py:90-92: DataLoader(shuffle=True) / an unseeded np.random.permutation -- the reference's order is not
        # reproducible and call_variants.sh sorts the records afterwards (:151); the set of sites is unchanged
        raise SystemExit("--shuffle_test is not supported: the reference's shuffled order is unseeded and the pipeline sorts "
                         "the scored records anyway (call_variants.sh:151); drop the flag")
    holdout = tuple(str(c) for c in (args.test_holdout_chromosomes or ()))
    # trainer.py:513-515: the loop breaks when batch > max_test_batches, batches of --test-batch-size sites
    site_limit = (args.max_test_batches + 1) * args.test_batch_size if args.max_test_batches > 0 else 0
    shard_i, shard_n = parse_shard(args.shard)
    if args.save_vcf_records:
        assert args.save_vcf_records_file != "", "Need a valid filename for args.save_vcf_records_file to save records"
    out_base = args.save_vcf_records_file or os.path.join(os.path.dirname(args.test_file), "model_test.vcf")
    out_final = scored_vcf_path(out_base)                     # <dir>/epoch1_<basename>, dl4vc/utils.py:152

    if args.gpus > 1 and not args.shard:
        # one process per GPU, contiguous shards, host-side concat
        t0 = time.time()
        procs = []
        devices = child_devices(args.gpus)
        for g in range(args.gpus):
            env = dict(os.environ, HIP_VISIBLE_DEVICES=devices[g], HSA_ENABLE_IPC_MODE_LEGACY="0")
            env.pop("CUDA_VISIBLE_DEVICES", None)             # (HIP honours both; the mask is carried in HIP_VISIBLE_DEVICES)
            cmd = [sys.executable, os.path.abspath(__file__)] + list(argv or sys.argv[1:]) + ["--shard", "%d/%d" % (g, args.gpus)]
            procs.append(subprocess.Popen(cmd, env=env))
        rcs = wait_children(procs)
        if any(rcs):
            for g in range(args.gpus):                        # no half-written parts left behind
                try:
                    os.remove(part_path(out_final, g))
                except OSError:
                    pass
            raise SystemExit("shard process failed: %s" % rcs)
        t_shards = time.time() - t0
        # what each shard did (its own scoring-loop clock, written beside its part file) and what the host-side concat costs: the
        # first run on a real multi-GPU node explains itself
        import json
        total = 0
        for g in range(args.gpus):
            side = part_path(out_final, g) + ".stats.json"
            try:
                with open(side) as f:
                    st = json.load(f)
                os.remove(side)
                total += st["sites"]
                print("\tshard %d/%d on device %s: %d sites, scoring loop %.2f s = %.0f sites/s (process %.2f s incl. start-up and "
                      "checkpoint load)" % (g, args.gpus, devices[g], st["sites"], st["loop_s"], st["sites"] / max(st["loop_s"], 1e-9),
                                            st["process_s"]))
            except (OSError, ValueError, KeyError):
                print("\tshard %d/%d: no statistics file" % (g, args.gpus))
        t1 = time.time()
        concat_parts(out_final, args.gpus, header_from=args.sample_vcf)
        t_cat = time.time() - t1
        print("\t%d shards: %d sites in %.2f s = %.0f sites/s whole job; host-side concat %.3f s" %
              (args.gpus, total, t_shards + t_cat, total / max(t_shards + t_cat, 1e-9), t_cat))
        print("\tTime elapsed for inference/testing {:.4f}".format(time.time() - t0))
        return 0

    from dl4vc_amd.model import DanNet, load_checkpoint
    from dl4vc_amd.inference import run_shard

    s_eval = time.time()
    print("Loading model checkpoint from {}".format(args.modelload))
    net = DanNet(cfg, device_id=0, max_batch=args.sites_per_launch).load_state_dict(load_checkpoint(args.modelload))
    if shard_n > 1:
        target = part_path(out_final, shard_i)
    else:
        target = out_final + ".records"
    t_loop = time.time()
    stats = None if os.environ.get("DL4VC_NO_THRESHOLD_STATS") else {}      # (the near-threshold count is a log line: opt out for raw rate)
    n = run_shard(net, args.test_file, target, shard_i, shard_n, sites_per_launch=args.sites_per_launch,
                  reads_seed=args.reads_seed, use_var_type_threshold=args.use_var_type_threshold,
                  holdout_chromosomes=holdout, site_limit=site_limit, log=lambda m: print(m, end="\r"), stats=stats)
    t_loop = time.time() - t_loop
    net.close()
    if stats is not None:
        print("\n%d of %d sites lie within 1e-4 of a genotype threshold of the published pipeline (format_vcf flags of "
              "call_variants.sh:154-160; main.py has no threshold flags of its own -- tools/format_vcf.py takes them later): only "
              "there could a call differ from another correct fp32 evaluation of the same scores"
              % (stats.get("near_threshold", 0), stats.get("sites", 0)))
    print("\nscoring loop (HDF5 read + assembly + forward + VCF text): %d sites in %.2f s = %.0f sites/s" % (n, t_loop, n / max(t_loop, 1e-9)))
    if shard_n == 1:
        if args.sample_vcf:
            start_scored_vcf(args.sample_vcf, out_base)
        else:
            open(out_final, "w").close()
        with open(out_final, "a") as out, open(target) as src:
            out.write(src.read())
        os.remove(target)
    if shard_n > 1:
        import json
        with open(target + ".stats.json", "w") as f:
            json.dump({"sites": int(n), "loop_s": t_loop, "process_s": time.time() - s_eval, "shard": shard_i, "of": shard_n}, f)
    print("\nscored %d sites -> %s" % (n, out_final if shard_n == 1 else target))
    print("\tTime elapsed for inference/testing {:.4f}".format(time.time() - s_eval))
    return 0


if __name__ == "__main__":
    sys.exit(main())
