# round 4, final state of the code: the whole GPU test suite, the driver's bench command (twice), the 200-step line, the line with an RCCL
# process group of one rank (the stream mode must not depend on it), and the training lines.  One box.   bash tools/evidence_r4_final.sh
R=$PWD; O=$R/gpurun_out/r4/final; mkdir -p $O
python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 > $O/bench_line.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 > $O/bench_line_second_run.json
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs 2>/dev/null | grep '^{' | tail -1 > $O/bench_line_200.json
VIDC_DIST_WORLD1=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | grep '^{' | tail -1 > $O/bench_line_rccl_world1.json
VIDC_TRAIN_PRECISION=bf16 python bench.py --train --batch 8 --steps 20 --warmup 3 2>/dev/null | grep '^{' | tail -1 > $O/train_line_bf16_b8.json
VIDC_TRAIN_PRECISION=fp32 python bench.py --train --batch 8 --steps 10 --warmup 3 2>/dev/null | grep '^{' | tail -1 > $O/train_line_fp32_b8.json
VIDC_DIST_WORLD1=1 VIDC_TRAIN_PRECISION=bf16 python bench.py --train --batch 8 --steps 10 --warmup 3 2>/dev/null | grep '^{' | tail -1 > $O/train_line_bf16_b8_rccl_world1_f32_buckets.json
VIDC_DIST_WORLD1=1 VIDC_TRAIN_GRAD_BF16=1 VIDC_TRAIN_PRECISION=bf16 python bench.py --train --batch 8 --steps 10 --warmup 3 2>/dev/null | grep '^{' | tail -1 > $O/train_line_bf16_b8_rccl_world1_bf16_buckets.json
VIDC_TRAIN_PRECISION=bf16 python tools/pack_bench.py 2>&1 | grep -v amdgpu > $O/pack_bench.txt
for f in $O/bench_line*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d['value'], d['dtype'], (d.get('conv_stack') or {}).get('at_measured_frame_rate',{}).get('frac_of_peak_executed'), d.get('value_mixed'), d.get('first_item_latency_ms'), [(name, e.get('value'), e.get('ms_per_step'), e.get('error')) for name, e in (d.get('extra_legs') or {}).items()])
PY
done
for f in $O/train_line*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'], d['losses'][-2:])
PY
done
cat $O/pack_bench.txt
