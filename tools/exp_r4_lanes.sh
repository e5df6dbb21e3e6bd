#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_frames_per_launch.py -x -q 2>&1 | tail -3
VIDC_PRECISION=fp32 python tools/group_timeline.py 20 3 2 > gpurun_out/r4_timeline_fp32_F2_L3_b.txt 2>&1
grep "=== rep" gpurun_out/r4_timeline_fp32_F2_L3_b.txt
for L in 2 3 4; do for K in 20 200; do
python bench.py --steps $K --warmup 5 --lanes $L --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lanes $L K $K: fp32', d['value'], d['conv_stack']['at_measured_frame_rate']['frac_of_peak_executed'], ' mixed', d['value_mixed'])"
done; done
