#!/bin/bash
# The first run on a multi-GPU node, self-explaining (reference: main.py:117 nn.DataParallel).
#
#   tools/scale_run.sh [--gpus N] [--sites S] [--out DIR] [--steps K]      on a node with N visible GPUs (default: all of them)
#   tools/scale_run.sh --rehearse-cpu N [--out DIR]                        no GPU: the same launch paths over gloo on the CPU
#
# For every n in {1, 2, 4, N}: `bench.py --gpus n` (DAN forward, site-sharded, no data-path collective) and `bench.py --mode train
# --gpus n` (one process per GPU, bucketed gradient exchange over RCCL), then `main.py --gpus N` on a generated S-site HDF5
# (default 1 048 576; contiguous shards, host-side concat).  One table at the end (tools/scale_table.py): sites/s per n, efficiency
# against n = 1, ranks_seen, every shard's own scoring-loop rate, the concat seconds, the gradient exchange's exposed milliseconds
# per step and which form of it ran.  Everything is kept under DIR (default gpurun_out/scale).
#
# The rehearsal replaces the GPU programs by tests/rehearse_scale_rank.py (the CPU oracle as the test double of the forward, the
# real GradientExchange / shard / concat / rank-counting code over gloo) and prints the same table from the same line schema: it
# proves the launch paths and the table, not a rate.
set -eo pipefail
cd "$(dirname "$0")/.."
gpus=""; sites=1048576; out=gpurun_out/scale; steps=5; rehearse=0; failed=0
while [ $# -gt 0 ]; do
    case "$1" in
        --gpus) gpus=$2; shift 2;;
        --sites) sites=$2; shift 2;;
        --out) out=$2; shift 2;;
        --steps) steps=$2; shift 2;;
        --rehearse-cpu) rehearse=$2; shift 2;;
        *) echo "unknown argument $1" >&2; exit 2;;
    esac
done
mkdir -p "$out"
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
if [ "$rehearse" -gt 0 ]; then
    N=$rehearse
else
    # (device_count does not initialise the GPU: this shell may still start rank processes)
    N=${gpus:-$(python -c 'import torch; print(torch.cuda.device_count())')}
    if [ "$N" -lt 1 ]; then echo "no GPU visible: use --rehearse-cpu N" >&2; exit 2; fi
fi
ns=$(python - "$N" <<'PY'
import sys
n = int(sys.argv[1])
print(" ".join(str(v) for v in sorted({v for v in (1, 2, 4, n) if v <= n})))
PY
)
echo "scale run: N = $N, ranks per run: $ns, out = $out" | tee "$out/scale_run.log"
port=$((20000 + RANDOM % 20000))
for n in $ns; do
    for mode in infer train; do
        f="$out/${mode}_n$n.json"
        echo "--- $mode, $n rank(s)" | tee -a "$out/scale_run.log"
        ok=1
        if [ "$rehearse" -gt 0 ]; then
            port=$((port + 1))
            python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$port" \
                tests/rehearse_scale_rank.py --mode "$mode" --gpus "$n" --steps 2 > "$f" 2> "$out/${mode}_n$n.err" || ok=0
        elif [ "$mode" = infer ]; then
            python bench.py --gpus "$n" --steps "$steps" --warmup 1 --no-cpu-baseline --no-skip-pass --no-host-path > "$f" 2> "$out/${mode}_n$n.err" || ok=0
        else
            python bench.py --mode train --gpus "$n" --steps $((steps * 4)) --warmup 3 --no-cpu-baseline > "$f" 2> "$out/${mode}_n$n.err" || ok=0
        fi
        # a run that fails must not take the table with it (set -e): its row reads FAILED there, the script exits non-zero at the end
        if [ "$ok" = 0 ]; then
            failed=$((failed + 1)); : > "$f"
            echo "run failed: see $out/${mode}_n$n.err" | tee -a "$out/scale_run.log"
        else
            tail -n 1 "$f" | cut -c1-200 | tee -a "$out/scale_run.log"
        fi
    done
done
# ---- the CLI path: main.py --gpus N on a generated candidates.hdf
echo "--- main.py --gpus $N on $sites generated sites" | tee -a "$out/scale_run.log"
ok=1
if [ "$rehearse" -gt 0 ]; then
    python tests/rehearse_scale_rank.py --mode cli --gpus "$N" --out "$out" > "$out/main_n$N.txt" 2> "$out/main_n$N.err" || ok=0
else
    python tools/scale_table.py --make-inputs "$out" --sites "$sites" &&
    python main.py --test_file "$out/candidates.hdf" --modelload "$out/ckpt.pth.tar" --sample_vcf "$out/candidates.vcf" --save_vcf_records \
        --save_vcf_records_file "$out/model_test.vcf" --gpus "$N" --sites-per-launch 4096 $(python tools/scale_table.py --model-flags) \
        > "$out/main_n$N.txt" 2> "$out/main_n$N.err" || ok=0
fi
if [ "$ok" = 0 ]; then
    failed=$((failed + 1)); rm -f "$out/main_n$N.txt"
    echo "run failed: see $out/main_n$N.err" | tee -a "$out/scale_run.log"
fi
python tools/scale_table.py --table "$out" --gpus "$N" | tee -a "$out/scale_run.log"
if [ "$failed" -gt 0 ]; then echo "$failed run(s) FAILED" | tee -a "$out/scale_run.log"; exit 1; fi
