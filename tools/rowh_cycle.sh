#!/bin/bash
# build + run tools/rowh_probe.hip on the GPU box (from the repository root): bash tools/rowh_cycle.sh [n_rows]
set -eo pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Idl4vc_amd/csrc -c tools/rowh_probe.hip -o /tmp/rowh_probe.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/rowh_probe.o dl4vc_amd/csrc/dan_train.o dl4vc_amd/csrc/dan_kernels.o -o /tmp/rowh_probe
timeout -k 10 200 /tmp/rowh_probe "$@" | tee gpurun_out/rowh_probe.txt
