#!/bin/bash
# build + run a probe on the GPU box: tools/rowh_probe.hip (default), or with "point" / "tails" as the first argument
# tools/point_probe.hip / tools/tails_probe.hip
set -eo pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
name=rowh_probe
objs="dl4vc_amd/csrc/dan_train.o dl4vc_amd/csrc/dan_kernels.o"
if [ "${1:-}" = point ]; then name=point_probe; shift; fi
if [ "${1:-}" = tails ]; then name=tails_probe; objs="dl4vc_amd/csrc/dan_kernels.o dl4vc_amd/csrc/dan_kernels_bf16p.o dl4vc_amd/csrc/dan_kernels_bf16x.o"; shift; fi
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Idl4vc_amd/csrc -c tools/$name.hip -o /tmp/$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/$name.o $objs -o /tmp/$name
timeout -k 10 200 /tmp/$name "$@" | tee gpurun_out/$name.txt
