# round 3: the training step's bf16 launches re-measured against all 39 tilings (the table dates from round 2's 27)
O=gpurun_out/r3/train_retune; mkdir -p $O
run() { VIDC_TRAIN_PRECISION=bf16 python tools/train_bench.py --batch 8 --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1:', d['ms_per_step'], 'ms per step', d['losses'][-1])"; }
run before; run before
cp vi_depth_completion_amd/train_tuning.json $O/train_tuning_before.json
timeout 2400 python tools/autotune_train.py --bf16 --remeasure > $O/autotune_train_bf16.log 2>&1; tail -2 $O/autotune_train_bf16.log
cp vi_depth_completion_amd/train_tuning.json $O/train_tuning_after.json
run after; run after
python -m pytest tests/test_training.py -q -x -m gpu 2>&1 | tail -2
