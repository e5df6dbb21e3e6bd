# round 3 experiment: 2-deep-ring tiles (32-48 KB of LDS, >= 3 workgroups per CU) swapped in for the table's choices, one to three lanes
python -m pytest tests/test_hip_parity.py -q -x -k "conv_tiles and (28- or 29- or 30- or 31- or 32-)" 2>&1 | tail -2
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sequential-leg --no-fp32-leg --lanes $1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VIDC_PRECISION remap=$VIDC_TILE_REMAP lanes $1:', d['value'], d['program_ms'])"; }
R1="6:28,10:28"
R2="6:28,10:28,4:29,7:30,11:30,21:32,22:32"
R3="6:28,10:28,4:29,7:30,11:30,21:32,22:32,3:31"
for prec in fp32 mixed; do
  export VIDC_PRECISION=$prec
  for remap in "" "$R1" "$R2" "$R3"; do
    export VIDC_TILE_REMAP=$remap
    for lanes in 1 2 3; do run $lanes; done
  done
done
