#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_frames_per_launch.py -x -q 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sequential-leg --no-extra-legs --per-op gpurun_out/r4_per_op_F2.tsv 2>gpurun_out/r4_perop.err | tail -1 > gpurun_out/r4_perop_line.json
python -c "
import json; d=json.load(open('gpurun_out/r4_perop_line.json')); print(d['value'], d['value_mixed'], d['conv_stack'], d['roofline'])"
