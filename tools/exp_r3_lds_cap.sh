mkdir -p gpurun_out/r3/exp1
export VIDC_PRECISION=fp32
for lanes in 2 3; do
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sequential-leg --lanes $lanes 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cap none lanes $lanes', d['value'], d['program_ms'])"
  for cap in 80 64; do
    VIDC_LDS_CAP_KB=$cap python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sequential-leg --lanes $lanes 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cap $cap lanes $lanes', d['value'], d['program_ms'])"
  done
done
