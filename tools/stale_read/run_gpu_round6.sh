#!/bin/bash
# stand-alone: the tagged-word consumers beside a noise kernel of another stream (MFMA bf16 / fp32, ds_read, LDS-DMA)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/stale
R=tools/stale_read/repro
M=gpurun_out/stale/matrix6.txt
: > $M
BASE="--lanes 3 --F 4 --items 1200 --nf0 10 --nf1 5"
for pat in stem stemlds band; do
  for nz in -1 0 1 2 4 5 6 8 9 10 13 14; do
    echo "## pattern $pat noise $nz" >> $M
    timeout 300 $R $BASE --pattern $pat --noise $nz --verbose 2>&1 | tail -12 >> $M || echo "   (exit $?)" >> $M
  done
done
grep -E "RESULT|words from" $M | cut -c1-260
