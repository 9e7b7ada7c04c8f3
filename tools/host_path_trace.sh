#!/bin/bash
# Timeline of the asynchronous host path (dan_forward_async / dan_wait) at BASELINE config 5's shape: kernel and memory-copy traces of
# one bench pass (no counters), condensed by tools/host_path_gaps.py into: copy bandwidths, GPU idle gaps, where the first launch starts.
#   gpurun -- tools/host_path_trace.sh [tag]
set -eo pipefail
tag=${1:-hp}
root=$PWD
out=$root/gpurun_out/trace_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$out/t" -- python3 "$root/bench.py" --precision 2 --reads 128 --window 301 --sites 16384 --steps 1 --warmup 1 --no-cpu-baseline --no-skip-pass --no-oracle-check > "$out/bench.json" 2> "$out/bench.err"
cd "$root"
python tools/host_path_gaps.py "$out/t" | tee "$out/gaps.txt"
