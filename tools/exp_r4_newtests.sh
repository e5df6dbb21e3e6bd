#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_checkpoints.py tests/test_frames_per_launch.py tests/test_configs.py "tests/test_training.py::test_two_ranks_through_training_steps" "tests/test_training.py::test_two_ranks_with_bf16_gradient_buckets" "tests/test_training.py::test_bf16_gradient_buckets_keep_the_loss_curve" tests/test_dorn.py -x -q -s 2>&1 | tail -40 > gpurun_out/r4_newtests.log
tail -30 gpurun_out/r4_newtests.log
python -m pytest tests/test_hip_parity.py -x -q -k "golden or demo" 2>&1 | tail -5
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err
tail -3 gpurun_out/r4_bench_default.err
python -c "
import json; d=json.loads(open('gpurun_out/r4_bench_default.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','dtype','ms_per_step','rmse_vs_oracle','value_mixed','rmse_vs_oracle_mixed','first_item_latency_ms']})
print(d['conv_stack']['at_measured_frame_rate'], d['config']['lanes'], d['mixed_leg']['lanes'])
print(json.dumps(d.get('extra_legs'), indent=1)[:3000])
print(d['cpu_baseline'])
"
