# round 3: re-measure the exact-fp32 configuration of the frame program's signatures (320x256, batch 1) with the 2-deep-ring tiles and a
# fine split-K grid, then the fp32 leg of bench.py with the new table
mkdir -p gpurun_out/r3/tune_fp32
python tools/autotune.py --heights 256 --batches 1 --splitk 1,2,3,4,5,6,7,8,10,12,16 --fp32-only --frame-only --verbose > gpurun_out/r3/tune_fp32/autotune.log 2>&1
cp vi_depth_completion_amd/conv_tuning.json gpurun_out/r3/tune_fp32/conv_tuning.json
grep "fp32 " gpurun_out/r3/tune_fp32/autotune.log | grep -v prec | awk '{ if ($3" "$4 != "(was "$7" "$8) print }' | head -80
export VIDC_PRECISION=fp32
for st in "20 5" "200 20"; do set -- $st
  python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new table, $1 steps:', d['value'], d['program_ms'])"
done
