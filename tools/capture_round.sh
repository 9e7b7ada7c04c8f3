#!/bin/bash
# Every measured artefact of a round, in two GPU-box calls (each fits gpurun's 20-minute limit):
#   tools/capture_round.sh r03 a      default bench under rocprofv3 (kernel trace, PMC passes) + the bench lines of the other configurations
#   tools/capture_round.sh r03 b      kernel traces and PMC passes of the bf16 config-5 command and of the training step
#   tools/capture_round.sh r05 c      (round 5) PMC passes of the 10-site training step, the tail / highway / bf16x3 probes, the host-path trace
# (then, here: python tools/summarize_profile.py r03 -- it also condenses the PMC passes of the bf16 and training commands)
set -eo pipefail
tag=${1:-r03}
part=${2:-all}
out=gpurun_out/lines_$tag
mkdir -p "$out"
if [ "$part" = a ] || [ "$part" = all ]; then
timeout -k 10 900 tools/profile_round.sh "$tag"
timeout -k 10 300 python bench.py --reads 100 --sites 32768 --steps 3 --warmup 1 --no-cpu-baseline > "$out/fp32_100x201.json" 2> "$out/fp32_100x201.err"
timeout -k 10 300 python bench.py --skip-empty-rows --steps 3 --warmup 1 --no-cpu-baseline > "$out/skip_empty_rows.json" 2> "$out/skip.err"
timeout -k 10 300 python bench.py --precision 1 --steps 3 --warmup 1 --no-cpu-baseline > "$out/bf16x3.json" 2> "$out/bf16x3.err"
timeout -k 10 300 python bench.py --precision 2 --reads 128 --window 301 --sites 16384 --steps 3 --warmup 1 --no-cpu-baseline > "$out/bf16_128x301.json" 2> "$out/bf16.err"
timeout -k 10 300 python bench.py --precision 2 --steps 3 --warmup 1 --no-cpu-baseline > "$out/bf16_64x201.json" 2> "$out/bf16b.err"
timeout -k 10 300 python bench.py --mode train --steps 10 --warmup 2 > "$out/train.json" 2> "$out/train.err"
timeout -k 10 300 python bench.py --mode train --train-batch 10 --steps 20 --warmup 3 --no-cpu-baseline > "$out/train_b10.json" 2> "$out/train_b10.err"
# windows of 301 columns on the two parity-grade precisions (two units per read; round 5)
timeout -k 10 300 python bench.py --reads 128 --window 301 --sites 8192 --steps 2 --warmup 1 --no-cpu-baseline --no-skip-pass > "$out/fp32_128x301.json" 2> "$out/fp32_128x301.err"
timeout -k 10 300 python bench.py --precision 1 --reads 128 --window 301 --sites 16384 --steps 3 --warmup 1 --no-cpu-baseline --no-skip-pass > "$out/bf16x3_128x301.json" 2> "$out/bf16x3_128x301.err"
fi
if [ "$part" = c ]; then
    timeout -k 10 400 tools/profile_pmc.sh "pmc_train_b10_$tag" --mode train --train-batch 10 --steps 2 --warmup 1 --no-cpu-baseline
    TRAIN_ARGS="--train-batch 10" timeout -k 10 300 tools/profile_train.sh "b10_$tag" > "$out/train_b10_kernels.txt" 2>&1
    bash tools/rowh_cycle.sh tails > /dev/null 2>&1 && cp gpurun_out/tails_probe.txt "$out/tails_probe.txt"
    hipcc -O3 --offload-arch=gfx950 tools/highway_probe.hip -o /tmp/highway_probe.bin 2> /dev/null && timeout -k 10 120 /tmp/highway_probe.bin > "$out/highway_probe.txt" 2>&1
    timeout -k 10 400 tools/host_path_trace.sh "$tag" > "$out/host_path_trace.txt" 2>&1
    echo captured part c; exit 0
fi
if [ "$part" = a ]; then echo captured part a; exit 0; fi
timeout -k 10 300 tools/profile_train.sh "$tag" > "$out/train_kernels.txt" 2>&1
# kernel trace of the bf16 config-5 command, then the PMC passes (own runs, counters only) of it and of the training step
root=$PWD
mkdir -p "gpurun_out/prof_bf16_$tag"
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/prof_bf16_$tag/kt" -- python3 "$root/bench.py" --precision 2 --reads 128 --window 301 --sites 4096 --steps 2 --warmup 1 --no-cpu-baseline --no-skip-pass --no-host-path > "$root/gpurun_out/prof_bf16_$tag/bench_under_rocprof.json" 2> "$root/gpurun_out/prof_bf16_$tag/kt.log" )
timeout -k 10 400 tools/profile_pmc.sh "pmc_bf16_$tag" --precision 2 --reads 128 --window 301 --sites 2048 --steps 1 --warmup 0 --no-cpu-baseline --no-skip-pass --no-host-path
timeout -k 10 400 tools/profile_pmc.sh "pmc_train_$tag" --mode train --steps 2 --warmup 1 --no-cpu-baseline
# ... and of the bf16x3 command (precision 1: dan_kernels_bf16x.hip)
mkdir -p "gpurun_out/prof_bf16x3_$tag"
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/prof_bf16x3_$tag/kt" -- python3 "$root/bench.py" --precision 1 --sites 16384 --steps 2 --warmup 1 --no-cpu-baseline --no-skip-pass --no-host-path > "$root/gpurun_out/prof_bf16x3_$tag/bench_under_rocprof.json" 2> "$root/gpurun_out/prof_bf16x3_$tag/kt.log" )
timeout -k 10 400 tools/profile_pmc.sh "pmc_bf16x3_$tag" --precision 1 --sites 4096 --steps 1 --warmup 0 --no-cpu-baseline --no-skip-pass --no-host-path
# stamped diagnostic build of the bf16 segment kernel and the GEMM-walk microbenchmarks (built here: binaries are not committed)
hipcc -O3 --offload-arch=gfx950 -DDAN_STAMPS tools/segp_probe.hip -o /tmp/segp_probe.bin 2> /dev/null
timeout -k 10 60 /tmp/segp_probe.bin 0 2 301 > "$out/segp_probe_segment1.txt" 2>&1
timeout -k 10 60 /tmp/segp_probe.bin 2 7 301 > "$out/segp_probe_segment2.txt" 2>&1
hipcc -O3 --offload-arch=gfx950 -DDAN_STAMPS tools/segx_probe.hip -o /tmp/segx_probe.bin 2> /dev/null
timeout -k 10 60 /tmp/segx_probe.bin 0 2 201 64 > "$out/segx_probe_segment1.txt" 2>&1
timeout -k 10 60 /tmp/segx_probe.bin 2 7 201 64 > "$out/segx_probe_segment2.txt" 2>&1
hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_bf16_feed.hip -o /tmp/mfma_bf16_feed.bin 2> /dev/null || true
hipcc -O3 --offload-arch=gfx950 tools/ubench/gemm_p_alone.hip -o /tmp/gemm_p_alone.bin 2> /dev/null || true
hipcc -O3 --offload-arch=gfx950 tools/ubench/gemm_p_lone.hip -o /tmp/gemm_p_lone.bin 2> /dev/null || true
# (the round-3 / round-4 microbenchmarks: kept for reference, not part of a round's record any more -- a failure here does not fail the capture)
( timeout -k 10 60 /tmp/mfma_bf16_feed.bin && timeout -k 10 60 /tmp/gemm_p_alone.bin && timeout -k 10 60 /tmp/gemm_p_lone.bin ) > "$out/ubench_mfma_bf16.txt" 2>&1 || echo "ubench skipped"
echo captured
