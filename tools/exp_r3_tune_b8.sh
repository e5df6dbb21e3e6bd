# round 3: batch-8 / batch-4 signatures (BASELINE configs[2], [3]: 320x240 after the device-side resize) re-measured with the pipelined
# tiles in the candidate set; bench before and after on the same box
mkdir -p gpurun_out/r3/tune_b8
b8() { python bench.py --batch 8 --source 640x480 --height 240 --steps 100 --warmup 10 --no-fp32-leg --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 b8 640x480:', d['value'], d['program_ms'], d['conv_stack']['at_measured_frame_rate'])"; }
b4() { python bench.py --batch 4 --source 1280x720 --height 240 --steps 100 --warmup 10 --no-fp32-leg --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 b4 1280x720:', d['value'], d['program_ms'], d['conv_stack']['at_measured_frame_rate'])"; }
b8 before; b4 before
python tools/autotune.py --heights 240 --batches 4,8 --splitk 1,2,3,4,8 --frame-only > gpurun_out/r3/tune_b8/autotune.log 2>&1
cp vi_depth_completion_amd/conv_tuning.json gpurun_out/r3/tune_b8/conv_tuning.json
tail -2 gpurun_out/r3/tune_b8/autotune.log
b8 after; b4 after
