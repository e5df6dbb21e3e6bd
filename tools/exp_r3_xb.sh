# round 3 A/B: the fp32 cross-barrier fragment prefetch (VIDC_FP32_XB=1 build, `make xb`) against the product library on one box
python -m pytest tests/test_hip_parity.py -q -x -k "conv_tiles or conv_splitk or conv_is_deterministic or pipelined_fragment or conv_splitk_shared" 2>&1 | tail -2
python -m pytest tests/test_training.py -q -x -k "plain_bf16_conv_mode" 2>&1 | tail -2
export VIDC_PRECISION=fp32
for lib in libvidc_xb.so libvidc.so libvidc_xb.so libvidc.so; do      # (make -C vi_depth_completion_amd/csrc xb first: libvidc_xb.so = with the prefetch)
  for st in "20 5" "200 20"; do set -- $st
    VIDC_LIB_NAME=$lib python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib fp32 $1 steps:', d['value'], d['program_ms'], 'rmse', d['rmse_vs_oracle'])"
  done
done
