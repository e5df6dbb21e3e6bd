#!/bin/bash
# round 4: frames_per_launch (pipeline.run_interleaved) -- tests, then the bench at F = 1, 2, 3, 4 and 20 / 200 steps
mkdir -p gpurun_out
python -m pytest tests/test_frames_per_launch.py "tests/test_configs.py::test_frame_output_does_not_depend_on_the_shard" -x -q 2>&1 | tail -15 > gpurun_out/r4_pairing_tests.log
cat gpurun_out/r4_pairing_tests.log
for F in 2 1 3 4; do
  for K in 20 200; do
    python bench.py --steps $K --warmup 5 --frames-per-launch $F --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>gpurun_out/r4_pairing_F${F}_K${K}.err | tail -1 > gpurun_out/r4_pairing_F${F}_K${K}.json
    python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r4_pairing_F${F}_K${K}.json"))
    print("F=${F} K=${K}: fp32 %.1f fps (frac %.4f, first item %.2f ms, tick %s)  mixed %.1f fps (first item %.2f ms)" % (
        d["value"], d["conv_stack"]["at_measured_frame_rate"]["frac_of_peak_executed"], d["first_item_latency_ms"], d["program_ms"],
        d["value_mixed"], d["mixed_leg"]["first_item_latency_ms"]))
except Exception as e:
    print("F=${F} K=${K}: failed", e)
PY
  done
done 2>&1 | tee gpurun_out/r4_pairing_sweep.txt
