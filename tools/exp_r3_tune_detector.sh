# round 3 experiment: the plane-mask detector's conv signatures at batch 8 re-measured against every tiling (incl. the round-3 ones),
# BASELINE configs[2] before and after on the same box
mkdir -p gpurun_out/r3/tune_det
run() { python bench.py --batch 8 --height 240 --source 640x480 --plane-head --steps 40 --warmup 5 --no-cpu-baseline --no-sequential-leg --no-fp32-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1:', d['value'], d['program_ms'])"; }
run before; run before
python tools/plane_mask_bench.py --batches 8 2>&1 | tail -3
cp vi_depth_completion_amd/conv_tuning.json gpurun_out/r3/tune_det/before.json
timeout 2400 python tools/autotune.py --detector --merge --heights 240 --batches 8 > gpurun_out/r3/tune_det/autotune.log 2>&1
tail -3 gpurun_out/r3/tune_det/autotune.log
cp vi_depth_completion_amd/conv_tuning.json gpurun_out/r3/tune_det/conv_tuning.json
run after; run after
python tools/plane_mask_bench.py --batches 8 2>&1 | tail -3
