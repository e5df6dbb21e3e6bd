# round 3: training, bf16: the BatchNorm backward behind a stride-1 conv without bias writes dY in its two bf16 forms only
python -m pytest tests/test_training.py -q -x -m gpu 2>&1 | tail -3
run() { python tools/train_bench.py --batch 8 --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VIDC_TRAIN_PRECISION skip_f32_dy=$VIDC_TRAIN_SKIP_F32_DY:', d['ms_per_step'], 'ms per step', d['losses'][-1])"; }
export VIDC_TRAIN_PRECISION=bf16
for rep in 1 2 3; do for f in 0 1; do export VIDC_TRAIN_SKIP_F32_DY=$f; run; done; done
