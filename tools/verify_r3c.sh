mkdir -p gpurun_out/r3/v4; O=gpurun_out/r3/v4
python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_line.json
VIDC_TRAIN_PRECISION=bf16 python bench.py --train --batch 8 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/train_line_bf16_b8.json
VIDC_TRAIN_PRECISION=fp32 python bench.py --train --batch 8 --steps 5 --warmup 3 2>/dev/null | tail -1 > $O/train_line_fp32_b8.json
for f in $O/bench_line.json $O/train_line_bf16_b8.json $O/train_line_fp32_b8.json; do python -c "import json,sys; d=json.loads(open('$f').read()); print('$f', d['value'], d.get('value_fp32'), d['ms_per_step'])"; done
