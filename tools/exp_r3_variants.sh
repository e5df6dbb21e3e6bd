mkdir -p gpurun_out/r3/exp5
python -m pytest tests/test_hip_parity.py -x -q -k "first_and_drain or lanes_are_bit_identical or interleaved_matches or interleaved_golden" 2>&1 | tail -3
for v in 1 0; do
  VIDC_TICK_VARIANTS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variants $v 20 steps:', d['value'], d['value_fp32'])"
done
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variants 1 200 steps:', d['value'], d['value_fp32'])"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sequential-leg --lanes 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variants 1 20 steps lanes 3:', d['value'], d['value_fp32'])"
