# round 3: Infinity-Cache weight prefetch (vidc_conv_desc.prefetch, engine links every conv to the next one's weights): A/B on one box
O=gpurun_out/r3/prefetch; mkdir -p $O
python -m pytest tests/test_hip_parity.py -q -x -k "prefetch" 2>&1 | tail -3
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sequential-leg --lanes $1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ahead=$VIDC_PREFETCH_AHEAD cap=$VIDC_PREFETCH_MB lanes $1:', d['value'], d['value_fp32'], d['program_ms'], d['fp32_leg']['program_ms'] if 'fp32_leg' in d else '')"; }
export VIDC_PREFETCH_MB=16
for rep in 1 2; do
for ahead in 0 1 2; do
  export VIDC_PREFETCH_AHEAD=$ahead
  run 1; run 2
done
done
export VIDC_PREFETCH_AHEAD=1
for cap in 6 48; do export VIDC_PREFETCH_MB=$cap; run 1; run 2; done
