#!/bin/bash
mkdir -p gpurun_out
VIDC_PRECISION=fp32 python tools/group_timeline.py 20 2 2 > gpurun_out/r4_timeline_fp32_F2_L2.txt 2>&1
VIDC_PRECISION=fp32 python tools/group_timeline.py 20 3 2 > gpurun_out/r4_timeline_fp32_F2_L3.txt 2>&1
for L in 3; do for K in 20 200; do
python bench.py --steps $K --warmup 5 --lanes $L --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lanes $L K $K:', d['value'], d['conv_stack']['at_measured_frame_rate']['frac_of_peak_executed'], d['value_mixed'])"
done; done
