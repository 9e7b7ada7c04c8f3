#!/bin/bash
# Regenerates the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repository root; tools/capture_round.sh calls it):
#   tools/profile_round.sh r01        -> gpurun_out/prof_r01/*, then `python tools/summarize_profile.py r01` writes profiles/
# Kernel trace of the default bench command; FETCH_SIZE / WRITE_SIZE / SQ counters in their own --pmc passes of a short
# bench (4096 sites = 32 chunks) -- counters are never combined with trace domains other than the kernel trace.
set -eo pipefail
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
root=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/kt" -- python3 "$root/bench.py" --no-cpu-baseline --no-skip-pass --no-host-path > "$root/$out/bench_under_rocprof.json" 2> "$root/$out/kt.log"
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$root/$out/$c" -- python3 "$root/bench.py" --sites 4096 --steps 1 --warmup 0 --no-cpu-baseline --no-skip-pass --no-host-path > "$root/$out/$c.json" 2> "$root/$out/$c.log"
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d "$root/$out/SQ" -- python3 "$root/bench.py" --sites 4096 --steps 1 --warmup 0 --no-cpu-baseline --no-skip-pass --no-host-path > "$root/$out/SQ.json" 2> "$root/$out/SQ.log"
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$root/$out/SQ2" -- python3 "$root/bench.py" --sites 4096 --steps 1 --warmup 0 --no-cpu-baseline --no-skip-pass --no-host-path > "$root/$out/SQ2.json" 2> "$root/$out/SQ2.log"
echo done
