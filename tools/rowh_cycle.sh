#!/bin/bash
# build + run tools/rowh_probe.hip (or, with "point" as the first argument, tools/point_probe.hip) on the GPU box
set -eo pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
name=rowh_probe
if [ "${1:-}" = point ]; then name=point_probe; shift; fi
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Idl4vc_amd/csrc -c tools/$name.hip -o /tmp/$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/$name.o dl4vc_amd/csrc/dan_train.o dl4vc_amd/csrc/dan_kernels.o -o /tmp/$name
timeout -k 10 200 /tmp/$name "$@" | tee gpurun_out/$name.txt
