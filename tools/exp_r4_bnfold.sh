#!/bin/bash
python -m pytest tests/test_training.py -x -q -k "folded_batchnorm or training_iteration_vs_reference or plain_bf16_training_mode or graph_replay" 2>&1 | tail -5
for rep in 1 2; do for F in 1 0; do
VIDC_TRAIN_BN_FOLD=$F VIDC_TRAIN_PRECISION=bf16 python bench.py --train --batch 8 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bf16 fold=$F:', d['ms_per_step'], d['value'], d['losses'][-2:])"
done; done
for F in 1 0; do
VIDC_TRAIN_BN_FOLD=$F VIDC_TRAIN_PRECISION=fp32 python bench.py --train --batch 8 --steps 5 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fp32 fold=$F:', d['ms_per_step'], d['value'], d['losses'][-2:])"
done
