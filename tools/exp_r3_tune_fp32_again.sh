# round 3: one more two-lane tuner pass over the fp32 entries (batch 1, 320x256), A/B of the table on the same box at 20 and 200 steps
O=gpurun_out/r3/tune_fp32c; mkdir -p $O
run() { python bench.py --steps $2 --warmup 5 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 steps $2:', d['value'], d['value_fp32'])"; }
cp vi_depth_completion_amd/conv_tuning.json $O/before.json
timeout 1500 python tools/autotune_lanes.py --height 256 --batch 1 --top 40 --budget-s 1100 --precision fp32 > $O/autotune_lanes_fp32.log 2>&1; grep -v amdgpu.ids $O/autotune_lanes_fp32.log | grep -v "> *\([0-9a-zA-Z]*\) *sk\([0-9]*\) .*\1 *sk\2 " | tail -12
cp vi_depth_completion_amd/conv_tuning.json $O/after.json
for rep in 1 2 3; do
  cp $O/before.json vi_depth_completion_amd/conv_tuning.json; run before 20; run before 200
  cp $O/after.json vi_depth_completion_amd/conv_tuning.json; run after 20; run after 200
done
