#!/bin/bash
# frames/s of the headline leg against the number of lanes (and the runtime's hardware-queue limit); steady-state rate beside it
cd "$(dirname "$0")/.."
for q in "" 8; do
for L in 2 3 4 5 6; do
  printf "queues %-3s lanes %d: " "${q:-def}" $L
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  python bench.py --lanes $L --steps 48 --warmup 12 --regions 3 --no-mixed-leg --no-extra-legs --no-sequential-leg --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('value'), d.get('regions'), 'steady', d.get('steady_state_frames_per_s'), 'first', d.get('first_item_latency_ms'))"
done; done
