#!/bin/bash
# experiment: HIP stream priorities per lane (does a favoured first lane shorten the fill of a 20-step region?)
cd "$(dirname "$0")/.."
for pr in "" "-1,0,0" "-1,-1,0" "-1,0,1" "0,0,-1"; do
  printf "priorities [%-8s]: " "$pr"
  VIDC_LANE_PRIORITIES="$pr" python bench.py --steps 20 --warmup 5 --no-mixed-leg --no-extra-legs --no-sequential-leg --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('value'), d.get('regions'), 'steady', d.get('steady_state_frames_per_s'), 'first', d.get('first_item_latency_ms'))"
done
