python -m pytest tests/test_hip_parity.py -q -x -k "(conv_bf16x3 and (33 or 34 or 35 or 36))" 2>&1 | tail -2
python tools/autotune.py --heights 256 --batches 1 --splitk 1,2 --sigs M5120_N768_K6912,M1280_N768_K6912 --verbose --dry 2>&1 | grep "prec 1"
python tools/autotune.py --heights 240 --batches 8 --splitk 1,2 --sigs M38400_N768_K6912,M9600_N768_K6912,M38400_N256_K2304 --verbose --dry 2>&1 | grep "prec 1"
