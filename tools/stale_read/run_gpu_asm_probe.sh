#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/stale
P=gpurun_out/stale/asm_probe.txt
: > $P
for rep in 1 2 3; do
timeout 300 python tools/stale_read/asm_delta.py --hsaco tools/stale_read/asm/base.hsaco --loads 2 --iters 600 --probe >> $P 2>&1 || echo "   (exit $?)" >> $P
done
cat $P | grep -v amdgpu.ids | cut -c1-200 | head -150
