#!/bin/bash
# PMC passes of tools/point_probe.hip (GPU box, from the repository root): per-dispatch counters of the pointwise launch kinds
set -eo pipefail
cd "$(dirname "$0")/.."
root=$PWD
mkdir -p gpurun_out/point_pmc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Idl4vc_amd/csrc -c tools/point_probe.hip -o /tmp/point_probe.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/point_probe.o dl4vc_amd/csrc/dan_train.o dl4vc_amd/csrc/dan_kernels.o -o /tmp/point_probe
export TMPDIR=/tmp
cd /tmp
pass() { rocprofv3 --pmc $2 --output-format csv -d "$root/gpurun_out/point_pmc/$1" -- /tmp/point_probe > "$root/gpurun_out/point_pmc/$1.txt" 2>&1; echo "pass $1 done"; }
pass SQ "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES"
pass SQ2 "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
pass SQ3 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
pass SQ4 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"
pass TC "FETCH_SIZE WRITE_SIZE"
