#!/bin/bash
# Round 5: measured table entries for the Winograd-domain GEMM launches and the per-layer choice direct / F(2x2) / F(4x4).
#   tools/tune_winograd.sh <batch> <height> <out dir>      (on the GPU box; writes the table in place and copies it to <out dir>)
set -u
B=${1:-4}; H=${2:-256}; OUT=${3:-gpurun_out/r5_tune}
mkdir -p $OUT
for m in 2 4; do
  VIDC_WINOGRAD=$m python tools/autotune.py --heights $H --batches $B --only-missing --merge --splitk 1,2,4 > $OUT/autotune_w${m}_b${B}_h${H}.log 2>&1
  tail -3 $OUT/autotune_w${m}_b${B}_h${H}.log
done
cp vi_depth_completion_amd/conv_tuning.json $OUT/conv_tuning_gemms.json
FPL=$B
for w in 0 2 4; do
  VIDC_WINOGRAD=$w python bench.py --height $H --frames-per-launch $FPL --steps 24 --warmup 12 --no-cpu-baseline --no-extra-legs --no-sequential-leg --per-op $OUT/per_op_w${w}_b${B}_h${H}.tsv > $OUT/bench_w${w}_b${B}_h${H}.json 2> $OUT/bench_w${w}_b${B}_h${H}.err
done
python tools/winograd_select.py --fp32 $OUT/per_op_w{0,2,4}_b${B}_h${H}.tsv.fp32 --mixed $OUT/per_op_w{0,2,4}_b${B}_h${H}.tsv > $OUT/select_b${B}_h${H}.log 2>&1
tail -40 $OUT/select_b${B}_h${H}.log
cp vi_depth_completion_amd/conv_tuning.json $OUT/conv_tuning.json
