#!/bin/bash
# Every measured artefact of a round in one GPU-box call: tools/capture_round.sh r02   (then, here: python tools/summarize_profile.py r02)
set -eo pipefail
tag=${1:-r02}
out=gpurun_out/lines_$tag
mkdir -p "$out"
timeout -k 10 900 tools/profile_round.sh "$tag"
timeout -k 10 300 python bench.py --reads 100 --sites 32768 --steps 3 --warmup 1 --no-cpu-baseline > "$out/fp32_100x201.json" 2> "$out/fp32_100x201.err"
timeout -k 10 300 python bench.py --skip-empty-rows --steps 3 --warmup 1 --no-cpu-baseline > "$out/skip_empty_rows.json" 2> "$out/skip.err"
timeout -k 10 300 python bench.py --precision 1 --steps 3 --warmup 1 --no-cpu-baseline > "$out/bf16x3.json" 2> "$out/bf16x3.err"
timeout -k 10 300 python bench.py --precision 2 --reads 128 --window 301 --sites 16384 --steps 3 --warmup 1 --no-cpu-baseline > "$out/bf16_128x301.json" 2> "$out/bf16.err"
timeout -k 10 300 python bench.py --precision 2 --steps 3 --warmup 1 --no-cpu-baseline > "$out/bf16_64x201.json" 2> "$out/bf16b.err"
timeout -k 10 300 python bench.py --mode train --steps 10 --warmup 2 > "$out/train.json" 2> "$out/train.err"
timeout -k 10 300 tools/profile_train.sh "$tag" > "$out/train_kernels.txt" 2>&1
echo captured
