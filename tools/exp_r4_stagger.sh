#!/bin/bash
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sequential-leg --no-extra-legs --no-mixed-leg > /dev/null 2>&1   # warm the box
for rep in 1 2 3 4; do for S in 0 2 1; do
VIDC_FILL_STAGGER=$S python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sequential-leg --no-extra-legs --no-mixed-leg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('stagger $S: fp32', d['value'], d['conv_stack']['at_measured_frame_rate']['frac_of_peak_executed'], d['first_item_latency_ms'])"
done; done
