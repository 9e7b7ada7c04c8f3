#!/bin/bash
# build + (on the GPU box) bf16x3 tests, stamped probe, short bench -- the inner loop of tuning dan_kernels_bf16x.hip
# usage (build container): tools/x3_cycle.sh build ; gpurun -- tools/x3_cycle.sh run [tag]
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
    make -C dl4vc_amd/csrc 2>&1 | grep -i "error" && exit 1
    hipcc -O3 --offload-arch=gfx950 -DDAN_STAMPS tools/segx_probe.hip -o tools/segx_probe.bin 2>&1 | grep -i error && exit 1
    exit 0
fi
tag=${2:-x}
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_hip_bf16.py -q -k "bf16x3" -x > gpurun_out/r4_${tag}_tests.txt 2>&1 || true
tail -2 gpurun_out/r4_${tag}_tests.txt
DAN_X_STAGGER=${STAG:-0} timeout -k 10 60 tools/segx_probe.bin 0 2 201 64 > gpurun_out/r4_${tag}_probe1.txt 2>&1
DAN_X_STAGGER=${STAG:-0} timeout -k 10 60 tools/segx_probe.bin 2 7 201 64 > gpurun_out/r4_${tag}_probe2.txt 2>&1
cat gpurun_out/r4_${tag}_probe1.txt gpurun_out/r4_${tag}_probe2.txt
timeout -k 10 300 python bench.py --precision 1 --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-skip-pass --no-oracle-check > gpurun_out/r4_${tag}_bench.txt 2>&1
tail -1 gpurun_out/r4_${tag}_bench.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('sites/s', d['value'], 'ms/step', d['ms_per_step'], 'segment ms/step', r.get('kernel_ms_per_step'), 'others', r.get('other_kernels_ms_per_step'))"
