# round 3: training step: weight packing with 64-unit blocks for 1x1 convs, 4-channel max-pool backward, 8-lane chunk reduction, unrolled
# stem weight-gradient loop -- tests, step time and the per-kernel table of one single-stream step
mkdir -p gpurun_out/r3/tail; O=$PWD/gpurun_out/r3/tail; R=$PWD
python -m pytest tests/test_training.py -q -x -m gpu 2>&1 | tail -3
run() { python tools/train_bench.py --batch 8 --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VIDC_TRAIN_PRECISION streams=$VIDC_TRAIN_STREAMS:', d['ms_per_step'], 'ms per step', d['losses'][-1])"; }
export VIDC_TRAIN_PRECISION=bf16
VIDC_TRAIN_STREAMS=3 run; VIDC_TRAIN_STREAMS=3 run; VIDC_TRAIN_STREAMS=1 run
VIDC_TRAIN_PRECISION=fp32 VIDC_TRAIN_STREAMS=3 run
cd /tmp && export TMPDIR=/tmp
VIDC_TRAIN_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 $R/tools/train_bench.py --batch 8 --steps 4 --warmup 4 > $O/train_line_prof.json 2> $O/prof.err
cp $(find $O/prof -name 't_kernel_stats.csv') $O/train_kernel_stats_bf16_1stream.csv; rm -rf $O/prof
head -30 $O/train_kernel_stats_bf16_1stream.csv | cut -c1-150
