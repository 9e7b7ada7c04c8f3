set -e
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/pmc_bf16_a
mkdir -p $out
cd /tmp
for form in p q; do
  export DAN_BF16_FORM=$form
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $out/sq1_$form -- $root/tools/segp_run.bin 2 7 301 > $out/sq1_$form.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $out/sq2_$form -- $root/tools/segp_run.bin 2 7 301 > $out/sq2_$form.log 2>&1 || true
done
unset DAN_BF16_FORM
find $out -name "*counter_collection.csv" | head
