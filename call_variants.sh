#!/bin/bash
# Drop-in for the reference's call_variants.sh from the candidate VCF on (call_variants.sh:85-168): candidates.vcf + BAM ->
# candidates.hdf (tools/convert_bam_single_reads.py here: own BAM / FASTA readers, no pysam), scoring with the MI355X-native DAN
# forward, then sort, genotype thresholds, multi-allele join and bgzip/tabix (with bcftools/htslib when installed, else in
# process: dl4vc_amd/vcfpost.py).  The first stage -- BAM -> candidates.vcf (tools/candidate_generator.py, reference
# call_variants.sh:76-83) -- is CPU pre-processing outside this implementation: OUTDIR must already hold candidates.vcf.
# With an OUTDIR that also holds candidates.hdf, -i / -r are not needed and the conversion is skipped.
set -e
usage() { echo "Usage: $0 -m MODEL -o OUTDIR [-i BAM -r REFERENCE] [-g GPUS] [-p PROCESSES]   (OUTDIR must hold candidates.vcf)"; exit 1; }
GPUS=1
PROCS=16
while getopts "m:o:g:i:r:b:p:h" opt; do
  case $opt in
    m) MODEL=$OPTARG ;;
    o) OUTDIR=$OPTARG ;;
    g) GPUS=$OPTARG ;;
    i) BAM=$OPTARG ;;
    r) REFERENCE=$OPTARG ;;
    b) BED=$OPTARG ;;       # (accepted for compatibility with the reference's flag line; only candidate generation uses it)
    p) PROCS=$OPTARG ;;
    *) usage ;;
  esac
done
[ -z "$MODEL" ] || [ -z "$OUTDIR" ] && usage
SCRIPTDIR="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
if [ ! -f "$OUTDIR/candidates.hdf" ]; then
  [ -f "$OUTDIR/candidates.vcf" ] && [ -n "$BAM" ] && [ -n "$REFERENCE" ] || { echo "missing $OUTDIR/candidates.hdf (or candidates.vcf with -i BAM -r REFERENCE to make it)"; exit 1; }
  printf "Convert candidates to HDF...\n"
  python "$SCRIPTDIR/tools/convert_bam_single_reads.py" --input "$BAM" --fp_vcf "$OUTDIR/candidates.vcf" \
      --fasta-input "$REFERENCE" --output "$OUTDIR/candidates.hdf" --max-reads 200 --num-processes "$PROCS" \
      --locations-process-step 100000 --max-insert-length 10 --max-insert-length-variant 50 \
      --save-q-scores --save-strand > "$OUTDIR/training_data.log" 2>&1
fi

printf "Run inference...\n"
python "$SCRIPTDIR/main.py" \
    --model-hidden-dropout 0.1 --model-batchnorm --model-use-q-scores --model-use-strands \
    --model-use-reads-ref-var-mask --model-conv-layers 7 --model-residual-layer-start 5 \
    --model-ave-pool-layers 2 --model-init-conv-channels 128 --model-final-conv-channels 128 \
    --model_pool_combine_dimension 0 --model-bottleneck-size 32 --model_final_layer_dilation 2 \
    --model_middle_layer_dilation 2 --model_concat_hw_reads --model-highway-single-reads \
    --gpus "$GPUS" --test-batch-size 200 --save_vcf_records \
    --save_vcf_records_file "$OUTDIR/model_test.vcf" --test_file "$OUTDIR/candidates.hdf" \
    --sample_vcf "$OUTDIR/candidates.vcf" --modelload "$MODEL" > "$OUTDIR/training.log" 2>&1

printf "Sort output VCF...\n"
awk '$1 ~ /^#/ {print $0;next} {print $0 | "sort -k1,1 -k2,2n"}' "$OUTDIR/epoch1_model_test.vcf" > "$OUTDIR/model_test_sorted.vcf"

printf "Threshold and combine multi-allele...\n"
python "$SCRIPTDIR/tools/format_vcf.py" --input_file "$OUTDIR/model_test_sorted.vcf" \
    --output_file "$OUTDIR/model_test_sorted_thres.vcf" --snp_threshold 0.1 --indel_threshold 0.2 \
    --snp_zygo_threshold 0.75 --indel_zygo_threshold 0.8 > "$OUTDIR/format_vcf.log" 2>&1

if command -v bcftools >/dev/null 2>&1; then
  bcftools norm -m +any "$OUTDIR/model_test_sorted_thres.vcf" > "$OUTDIR/model_test_sorted_thres-join.vcf" 2> "$OUTDIR/bcftools_norm.log"
  sed -i 's/0\/2/1\/2/' "$OUTDIR/model_test_sorted_thres-join.vcf"
  sed -i 's/2\/2/1\/2/' "$OUTDIR/model_test_sorted_thres-join.vcf"
  bgzip -c "$OUTDIR/model_test_sorted_thres-join.vcf" > "$OUTDIR/called_variants.vcf.gz"
  tabix -p vcf "$OUTDIR/called_variants.vcf.gz"
  echo "Called variants in $OUTDIR/called_variants.vcf.gz"
else
  # no bcftools / bgzip / tabix on this machine: the in-process restatement of the same five commands
  python "$SCRIPTDIR/tools/finish_calls.py" --input_file "$OUTDIR/model_test_sorted_thres.vcf" \
      --joined_file "$OUTDIR/model_test_sorted_thres-join.vcf" --output_gz "$OUTDIR/called_variants.vcf.gz"
  echo "Called variants in $OUTDIR/called_variants.vcf.gz (joined and indexed in process: bcftools not found)"
fi
