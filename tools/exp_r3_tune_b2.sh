# round 3: batch 2 at 320x256 was never measured (cost-model plan): isolated pass for the missing signatures, then the two-lane tuner
O=gpurun_out/r3/tune_b2; mkdir -p $O
run() { python bench.py --batch 2 --steps 100 --warmup 10 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1:', d['value'], d['value_fp32'], d['program_ms'])"; }
run before
timeout 1500 python tools/autotune.py --heights 256 --batches 2 --only-missing > $O/autotune_isolated.log 2>&1; tail -2 $O/autotune_isolated.log
run isolated
cp vi_depth_completion_amd/conv_tuning.json $O/conv_tuning_isolated.json
timeout 1500 python tools/autotune_lanes.py --height 256 --batch 2 --top 12 --budget-s 900 > $O/autotune_lanes.log 2>&1; tail -3 $O/autotune_lanes.log
run lanes_mixed
cp vi_depth_completion_amd/conv_tuning.json $O/conv_tuning_lanes.json
timeout 1500 python tools/autotune_lanes.py --height 256 --batch 2 --top 10 --budget-s 700 --precision fp32 > $O/autotune_lanes_fp32.log 2>&1; tail -3 $O/autotune_lanes_fp32.log
run lanes_fp32
cp vi_depth_completion_amd/conv_tuning.json $O/conv_tuning_final.json
