# round 3: training: chunk sums of the BatchNorm reductions added by the last workgroup of the partial-sum launch (670 launches fewer per step)
python -m pytest tests/test_training.py -q -x -m gpu 2>&1 | tail -3
run() { python tools/train_bench.py --batch 8 --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VIDC_TRAIN_PRECISION bn_tickets=$VIDC_TRAIN_BN_TICKETS streams=$VIDC_TRAIN_STREAMS:', d['ms_per_step'], 'ms per step', d['losses'][-1])"; }
export VIDC_TRAIN_PRECISION=bf16 VIDC_TRAIN_STREAMS=3
for rep in 1 2 3; do for f in 0 1; do export VIDC_TRAIN_BN_TICKETS=$f; run; done; done
export VIDC_TRAIN_STREAMS=1
for f in 0 1; do export VIDC_TRAIN_BN_TICKETS=$f; run; done
export VIDC_TRAIN_PRECISION=fp32 VIDC_TRAIN_STREAMS=3
for f in 0 1; do export VIDC_TRAIN_BN_TICKETS=$f; run; done
