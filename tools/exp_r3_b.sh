mkdir -p gpurun_out/r3/expb
python -m pytest tests/test_hip_parity.py -x -q -k "lanes_are_bit_identical or first_and_drain or interleaved_matches" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3/expb/bench20.json
python -c "import json; d=json.load(open('gpurun_out/r3/expb/bench20.json')); print('20 steps:', d['value'], d['value_fp32'], d['sequential_call_cnn'], d['fp32_leg']['sequential_call_cnn'])"
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('200 steps:', d['value'], d['value_fp32'])"
R2="6:28,10:28,4:29,7:30,11:30,21:32,22:32"
for remap in "25:1,24:1" "25:1,24:1,$R2" "25:1,24:1,2:29,20:29,17:29,19:31,3:31,$R2"; do
  for lanes in 2 3; do
    VIDC_TILE_REMAP=$remap python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sequential-leg --no-fp32-leg --lanes $lanes 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mixed remap=$remap lanes $lanes:', d['value'], d['program_ms'])"
  done
done
VIDC_TRAIN_PRECISION=bf16 python bench.py --train --batch 8 --steps 5 --warmup 3 2>/dev/null | tail -1 | cut -c1-600
python tools/dump_config2_detections.py gpurun_out/r3/config2_det.npz 2>&1 | tail -1
