#!/bin/bash
# runs tools/stale_read/asm_delta.py over the code objects given as "name:loads" pairs
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/stale
P=gpurun_out/stale/asm_delta.txt
for spec in "$@"; do
  n=${spec%%:*}; l=${spec##*:}
  timeout 300 python tools/stale_read/asm_delta.py --hsaco tools/stale_read/asm/$n.hsaco --loads $l --iters 1000 >> $P 2>&1 || echo "   ($spec exit $?)" >> $P
done
grep -E "ASM|exit|rror|ssert" $P | tail -40
