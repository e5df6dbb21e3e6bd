# round 3: the plane block of a lane on a side stream of the lane (beside segment 1) -- A/B on one box
mkdir -p gpurun_out/r3
python -m pytest tests/test_hip_parity.py tests/test_configs.py -q -x -k "interleaved or lanes or config or golden or dense" 2>&1 | tail -3
run() { python bench.py --steps $2 --warmup 10 --no-cpu-baseline --no-sequential-leg --lanes 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('side=$VIDC_PLANE_SIDE_STREAM steps $2:', d['value'], d['value_fp32'])"; }
for rep in 1 2 3; do
  for side in 0 1; do export VIDC_PLANE_SIDE_STREAM=$side; run 2 200; run 2 20; done
done
for side in 0 1; do export VIDC_PLANE_SIDE_STREAM=$side
python bench.py --batch 8 --source 640x480 --height 240 --plane-head --steps 60 --warmup 10 --no-fp32-leg --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('side=$VIDC_PLANE_SIDE_STREAM configs[2]:', d['value'])"
python bench.py --batch 4 --source 1280x720 --height 240 --steps 60 --warmup 10 --no-fp32-leg --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('side=$VIDC_PLANE_SIDE_STREAM configs[3]:', d['value'])"
done
