mkdir -p gpurun_out/r4/tune_fp32_b2
O=gpurun_out/r4/tune_fp32_b2
cp vi_depth_completion_amd/conv_tuning.json $O/before.json
timeout 2700 python tools/autotune.py --heights 256 --batches 2 --splitk 1,2,3,4,5,6,7,8,10,12,16 --fp32-only --frame-only --verbose --out $O/after.json > $O/autotune.log 2>&1
grep "fp32 " $O/autotune.log | grep -v prec | awk '{ if ($3" "$4 != "(was "$7" "$8) print }' | head -60
run() { python bench.py --steps $2 --warmup 5 --no-cpu-baseline --no-sequential-leg --no-extra-legs --no-mixed-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 steps $2: fp32', d['value'], d['program_ms'])"; }
if [ -f $O/after.json ]; then
for rep in 1 2; do
  cp $O/before.json vi_depth_completion_amd/conv_tuning.json; run before 20; run before 200
  cp $O/after.json vi_depth_completion_amd/conv_tuning.json; run after 20; run after 200
done
fi
cp $O/before.json vi_depth_completion_amd/conv_tuning.json
