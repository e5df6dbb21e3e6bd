# round 3: timeline of one captured training step (bf16, batch 8): concurrency, per-queue gaps.  Also: 3 lanes at the driver's 20 steps.
R=$PWD; O=$R/gpurun_out/r3/tl; mkdir -p $O
for lanes in 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sequential-leg --lanes $lanes 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('20 steps lanes $lanes:', d['value'], d['value_fp32'])"; done
for lanes in 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sequential-leg --lanes $lanes 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('20 steps lanes $lanes:', d['value'], d['value_fp32'])"; done
cd /tmp && export TMPDIR=/tmp
for streams in 3 1; do
  VIDC_TRAIN_STREAMS=$streams VIDC_TRAIN_PRECISION=bf16 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$streams -o t -- python3 $R/tools/train_bench.py --batch 8 --steps 4 --warmup 4 > $O/train_line_$streams.json 2> $O/prof_$streams.err
  python $R/tools/train_timeline.py $(find $O/prof_$streams -name 't_kernel_trace.csv') > $O/timeline_streams$streams.txt 2>&1
  rm -rf $O/prof_$streams
  tail -1 $O/train_line_$streams.json | cut -c1-200; cat $O/timeline_streams$streams.txt
done
