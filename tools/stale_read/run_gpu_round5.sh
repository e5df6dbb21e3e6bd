#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/stale
P=gpurun_out/stale/noise5.txt
: > $P
for nz in none conv_fp32 conv_bf16x3 conv_fp32_splitout conv_bf16x3_splitout split stem_f32 stem_split maxpool_f32 maxpool_split wino_in wino_in_split memcpy fill; do
  VIDC_DBG_STEM_LOADS=3 timeout 300 python tools/stale_read/noise_bisect.py --noise $nz --iters 400 >> $P 2>&1 || echo "   ($nz exit $?)" >> $P
done
grep -E "NOISE|exit|rror" $P
