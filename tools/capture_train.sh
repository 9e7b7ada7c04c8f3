#!/bin/bash
# The training step's measured artefacts alone (GPU box, from the repository root): tools/capture_train.sh r04
# -> gpurun_out/lines_<tag>/train*.json, gpurun_out/prof_train_<tag>/, gpurun_out/pmc_train_<tag>/ (then: python tools/summarize_profile.py <tag>)
set -eo pipefail
tag=${1:-r04}
out=gpurun_out/lines_$tag
mkdir -p "$out"
timeout -k 10 300 python bench.py --mode train --steps 10 --warmup 2 > "$out/train.json" 2> "$out/train.err"
timeout -k 10 300 python bench.py --mode train --train-batch 10 --steps 20 --warmup 3 --no-cpu-baseline > "$out/train_b10.json" 2> "$out/train_b10.err"
timeout -k 10 300 tools/profile_train.sh "$tag" > "$out/train_kernels.txt" 2>&1
timeout -k 10 400 tools/profile_pmc.sh "pmc_train_$tag" --mode train --steps 2 --warmup 1 --no-cpu-baseline
echo captured
