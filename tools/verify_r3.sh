mkdir -p gpurun_out/r3/v2
O=gpurun_out/r3/v2
python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > $O/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_line.json
python bench.py --batch 8 --source 640x480 --height 240 --steps 100 --warmup 10 --no-fp32-leg --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 > $O/bench_line_b8_640x480.json
python bench.py --batch 8 --source 640x480 --height 240 --plane-head --steps 100 --warmup 10 --no-fp32-leg --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 > $O/bench_line_b8_640x480_plane_head.json
python bench.py --batch 4 --source 1280x720 --height 240 --steps 100 --warmup 10 --no-fp32-leg --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 > $O/bench_line_b4_1280x720.json
cat $O/pytest_gpu.log; for f in $O/bench_line*.json; do python -c "import json,sys; d=json.loads(open('$f').read()); print('$f', d['value'], d.get('value_fp32'), d['roofline']['traffic'])"; done
