#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/stale
P=gpurun_out/stale/noise11.txt
: > $P
for loads in 58 66 58 66; do
  VIDC_DBG_STEM_LOADS=$loads timeout 300 python tools/stale_read/noise_bisect.py --noise conv_bf16x3 --iters 1000 >> $P 2>&1 || echo "   (loads $loads exit $?)" >> $P
done
for loads in 58 66; do
  VIDC_DBG_STEM_LOADS=$loads timeout 300 python tools/stale_read/noise_bisect.py --noise none --iters 1000 >> $P 2>&1
  VIDC_DBG_STEM_LOADS=$loads timeout 300 python tools/stale_read/noise_bisect.py --noise conv_fp32 --iters 1000 >> $P 2>&1
done
grep -E "NOISE|exit|rror" $P
