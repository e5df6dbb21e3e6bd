# round 3: two-lane tuner at batch 8 (configs[2]) over the 24 most expensive signatures with every tiling; A/B of the table on the same box
O=gpurun_out/r3/tune_b8l; mkdir -p $O
run() { python bench.py --batch 8 --height 240 --source 640x480 --steps 60 --warmup 8 --no-cpu-baseline --no-sequential-leg --no-fp32-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1:', d['value'], d['program_ms'])"; }
cp vi_depth_completion_amd/conv_tuning.json $O/before.json
run before; run before
timeout 2000 python tools/autotune_lanes.py --height 240 --batch 8 --top 24 --budget-s 1500 > $O/autotune_lanes_b8.log 2>&1; grep -v amdgpu.ids $O/autotune_lanes_b8.log | tail -30
cp vi_depth_completion_amd/conv_tuning.json $O/after.json
run after; run after
cp $O/before.json vi_depth_completion_amd/conv_tuning.json; run before_again
cp $O/after.json vi_depth_completion_amd/conv_tuning.json; run after_again
