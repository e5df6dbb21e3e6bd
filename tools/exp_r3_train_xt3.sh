# round 3: training, bf16: the 3x3 / strided convs' wgrad right operand gathered from the bf16 operand copy as well
python -m pytest tests/test_training.py -q -x -m gpu 2>&1 | tail -3
run() { python tools/train_bench.py --batch 8 --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VIDC_TRAIN_PRECISION xt_bf16=$VIDC_TRAIN_XT_BF16:', d['ms_per_step'], 'ms per step', d['losses'][-1])"; }
export VIDC_TRAIN_PRECISION=bf16
for rep in 1 2 3; do for f in 0 1; do export VIDC_TRAIN_XT_BF16=$f; run; done; done
