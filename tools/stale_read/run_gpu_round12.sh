#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/stale
P=gpurun_out/stale/noise12.txt
: > $P
for nz in conv_bf16x3 conv_bf16x3 none conv_fp32; do
  VIDC_DBG_STEM_LOADS=74 timeout 300 python tools/stale_read/noise_bisect.py --noise $nz --iters 1000 >> $P 2>&1 || echo "   (exit $?)" >> $P
done
grep -E "NOISE|exit|rror" $P
