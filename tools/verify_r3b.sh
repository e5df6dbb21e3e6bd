mkdir -p gpurun_out/r3/v3; O=gpurun_out/r3/v3
python tools/weight_warmth_probe.py --top 14 > $O/warmth_mixed.txt 2>&1
python tools/weight_warmth_probe.py --top 14 --precision fp32 > $O/warmth_fp32.txt 2>&1
cat $O/warmth_mixed.txt $O/warmth_fp32.txt
python bench.py --batch 2 --steps 100 --warmup 10 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 > $O/bench_line_batch2.json
python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
