#!/bin/bash
# PMC passes of one bench.py command (run on the GPU box from the repository root):
#   tools/profile_pmc.sh <out-name> <bench.py arguments...>
# -> gpurun_out/<out-name>/{FETCH_SIZE,WRITE_SIZE,SQ,SQ2,SQ3}/**/counter_collection.csv; `python tools/summarize_pmc.py` condenses them.
# Counters are collected in their own runs (no trace domain besides the implicit kernel dispatch records), one small group per pass.
set -eo pipefail
name=$1; shift
root=$PWD
out=$root/gpurun_out/$name
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
pass() {
    local d=$1; shift
    local ctrs=$1; shift
    rocprofv3 --pmc $ctrs --output-format csv -d "$out/$d" -- python3 "$root/bench.py" "$@" > "$out/$d.json" 2> "$out/$d.log"
    echo "pass $d done"
}
pass FETCH_SIZE "FETCH_SIZE" "$@"
pass WRITE_SIZE "WRITE_SIZE" "$@"
pass SQ "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "$@"
pass SQ2 "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "$@"
pass SQ3 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "$@"
