#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/stale
P=gpurun_out/stale/noise10.txt
: > $P
for loads in 2 42 50 42 50; do
  VIDC_DBG_STEM_LOADS=$loads timeout 300 python tools/stale_read/noise_bisect.py --noise conv_bf16x3 --iters 1000 >> $P 2>&1 || echo "   (loads $loads exit $?)" >> $P
done
grep -E "NOISE|exit|rror" $P
