# round 3: at the start of a stream a lane takes its second frame as soon as its first is enriched (before the next lane's hypothesis draws)
python -m pytest tests/test_hip_parity.py tests/test_configs.py tests/test_dorn.py -q -x -k "interleaved or lanes or config or golden or variants" 2>&1 | tail -3
run() { python bench.py --steps $1 --warmup 5 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('early=$VIDC_EARLY_SECOND_FRAME steps $1:', d['value'], d['value_fp32'])"; }
for rep in 1 2 3 4; do for e in 0 1; do export VIDC_EARLY_SECOND_FRAME=$e; run 20; done; done
for e in 0 1; do export VIDC_EARLY_SECOND_FRAME=$e; run 200; done
