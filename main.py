#!/usr/bin/env python3
"""Drop-in for the inference-only mode of the reference's ``main.py`` (main.py:47-229, branch :213-222).

    python main.py <the flags call_variants.sh passes> --test_file X.hdf --modelload CKPT \
        --save_vcf_records --save_vcf_records_file OUT/model_test.vcf --sample_vcf OUT/candidates.vcf

Reads the candidate HDF5 (schema of tools/convert_bam_single_reads.py), scores every site with the
MI355X-native DAN forward and writes ``OUT/epoch1_model_test.vcf`` exactly where and how the reference does
(dl4vc/utils.py:146-178).  ``--gpus N`` starts one process per GPU over contiguous site shards and
concatenates the part files on the host; there is no collective on this path.  Training flags are parsed
(the pipeline script passes them) and ignored; ``--train_file`` is rejected: training is out of scope.
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


def main(argv=None) -> int:
    args = create_arg_parser().parse_args(argv)
    print(args)
    if args.train_file:
        raise SystemExit("training (--train_file) is outside this implementation's scope (inference hot path only)")
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
        rcs = [p.wait() for p in procs]
        if any(rcs):
            for g in range(args.gpus):                        # no half-written parts left behind
                try:
                    os.remove(part_path(out_final, g))
                except OSError:
                    pass
            raise SystemExit("shard process failed: %s" % rcs)
        concat_parts(out_final, args.gpus, header_from=args.sample_vcf)
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
    n = run_shard(net, args.test_file, target, shard_i, shard_n, sites_per_launch=args.sites_per_launch,
                  reads_seed=args.reads_seed, use_var_type_threshold=args.use_var_type_threshold,
                  holdout_chromosomes=holdout, site_limit=site_limit, log=lambda m: print(m, end="\r"))
    net.close()
    if shard_n == 1:
        if args.sample_vcf:
            start_scored_vcf(args.sample_vcf, out_base)
        else:
            open(out_final, "w").close()
        with open(out_final, "a") as out, open(target) as src:
            out.write(src.read())
        os.remove(target)
    print("\nscored %d sites -> %s" % (n, out_final if shard_n == 1 else target))
    print("\tTime elapsed for inference/testing {:.4f}".format(time.time() - s_eval))
    return 0


if __name__ == "__main__":
    sys.exit(main())
