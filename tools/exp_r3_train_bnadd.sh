# round 3: training: the Bottleneck tail relu(bn3(.) + identity) inside bn3's apply pass
python -m pytest tests/test_training.py -q -x -m gpu 2>&1 | tail -3
run() { python tools/train_bench.py --batch 8 --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VIDC_TRAIN_PRECISION bn_add_fused=$VIDC_TRAIN_BN_ADD_FUSED:', d['ms_per_step'], 'ms per step', d['losses'][-1])"; }
export VIDC_TRAIN_PRECISION=bf16
for rep in 1 2 3; do for f in 0 1; do export VIDC_TRAIN_BN_ADD_FUSED=$f; run; done; done
export VIDC_TRAIN_PRECISION=fp32
for f in 0 1; do export VIDC_TRAIN_BN_ADD_FUSED=$f; run; done
