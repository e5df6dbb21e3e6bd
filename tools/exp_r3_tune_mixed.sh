# round 3: the frame program's signatures (320x256, batch 1) re-measured in both arithmetic modes with the 2-deep-ring tiles and a fine
# split-K grid; then bench.py with the new table (kept only if the two-lane stream is faster)
mkdir -p gpurun_out/r3/tune_mixed
cp vi_depth_completion_amd/conv_tuning.json gpurun_out/r3/tune_mixed/conv_tuning_before.json
python tools/autotune.py --heights 256 --batches 1 --splitk 1,2,3,4,5,6,7,8,10,12,16 --frame-only > gpurun_out/r3/tune_mixed/autotune.log 2>&1
cp vi_depth_completion_amd/conv_tuning.json gpurun_out/r3/tune_mixed/conv_tuning.json
tail -3 gpurun_out/r3/tune_mixed/autotune.log
for st in "20 5" "200 20"; do set -- $st
  python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new table, $1 steps:', d['value'], d['value_fp32'], d['program_ms'])"
done
cp gpurun_out/r3/tune_mixed/conv_tuning_before.json vi_depth_completion_amd/conv_tuning.json
for st in "20 5" "200 20"; do set -- $st
  python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old table, $1 steps:', d['value'], d['value_fp32'], d['program_ms'])"
done
