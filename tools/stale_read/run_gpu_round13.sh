#!/bin/bash
# after the rebuild without SLP-vectorised packed fp32: the control form must still fail, every other form of the fused stem and the default warp + stem pair must not
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/stale
P=gpurun_out/stale/noise13.txt
: > $P
for loads in 3 2 7 0; do
  VIDC_DBG_STEM_LOADS=$loads timeout 300 python tools/stale_read/noise_bisect.py --noise conv_bf16x3 --iters 1000 >> $P 2>&1 || echo "   (loads $loads exit $?)" >> $P
done
VIDC_DBG_STEM_LOADS=0 timeout 300 python tools/stale_read/noise_bisect.py --noise conv_bf16x3 --iters 1000 --victim warp >> $P 2>&1
timeout 120 tools/stale_read/pkmul 1 10 5 >> $P 2>&1
timeout 120 tools/stale_read/pkmul 5 10 5 >> $P 2>&1
grep -E "NOISE|PKMUL|exit|rror" $P
