#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/stale
P=gpurun_out/stale/noise8.txt
: > $P
for loads in 3 2 10 18 26 34 0 1; do
  VIDC_DBG_STEM_LOADS=$loads timeout 300 python tools/stale_read/noise_bisect.py --noise conv_bf16x3 --iters 800 >> $P 2>&1 || echo "   (loads $loads exit $?)" >> $P
done
VIDC_DBG_STEM_LOADS=0 timeout 300 python tools/stale_read/noise_bisect.py --noise conv_bf16x3 --iters 2000 --victim warp >> $P 2>&1
VIDC_DBG_STEM_LOADS=3 timeout 300 python tools/stale_read/noise_bisect.py --noise conv_bf16x3 --iters 800 --victims 1 >> $P 2>&1
VIDC_DBG_STEM_LOADS=3 GPU_MAX_HW_QUEUES=1 timeout 300 python tools/stale_read/noise_bisect.py --noise conv_bf16x3 --iters 800 >> $P 2>&1
grep -E "NOISE|exit|rror" $P
