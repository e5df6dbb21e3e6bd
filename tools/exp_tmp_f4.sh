mkdir -p gpurun_out
timeout 2400 python tools/autotune.py --heights 256 --batches 4 --only-missing --frame-only --out gpurun_out/conv_tuning_f4.json > gpurun_out/r4_autotune_b4_256.log 2>&1
tail -3 gpurun_out/r4_autotune_b4_256.log
cp gpurun_out/conv_tuning_f4.json vi_depth_completion_amd/conv_tuning.json
for L in 2 3; do for K in 20 200 400; do
python bench.py --steps $K --warmup 8 --lanes $L --frames-per-launch 4 --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('F=4 lanes $L K $K: fp32', d['value'], d['conv_stack']['at_measured_frame_rate']['frac_of_peak_executed'], d['program_ms'], d['first_item_latency_ms'], ' mixed', d['value_mixed'], d['mixed_leg']['program_ms'])"
done; done
