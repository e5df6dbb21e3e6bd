#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/stale
P=gpurun_out/stale/pkmul2.txt
: > $P
for form in 0 9 3 4 5 6 7 8 1 2; do timeout 120 tools/stale_read/pkmul 1 $form 5 >> $P 2>&1; done
for nz in 4 5 2; do timeout 120 tools/stale_read/pkmul $nz 0 5 >> $P 2>&1; done
cat $P
