set -x
R=$PWD; O=$R/gpurun_out/r2/final; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_line.json
python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_200.json
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-fp32-leg --lanes 1 2>/dev/null | tail -1 > $O/bench_line_200_lanes1.json
python bench.py --batch 8 --source 640x480 --height 240 --steps 100 --warmup 10 --no-fp32-leg --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_b8_640x480.json
python bench.py --batch 8 --source 640x480 --height 240 --plane-head --steps 100 --warmup 10 --no-fp32-leg --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_b8_640x480_plane_head.json
python bench.py --batch 4 --source 1280x720 --height 240 --steps 100 --warmup 10 --no-fp32-leg --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_b4_1280x720.json
python bench.py --plane-head --steps 200 --warmup 20 --no-fp32-leg --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_plane_head.json
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-fp32-leg --per-op $O/per_op.tsv 2>/dev/null | tail -1 > /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -o r2 -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-fp32-leg > $O/bench_line_profiled.json 2> $O/prof2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o r2 -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-fp32-leg --lanes 1 > $O/bench_line_profiled_lanes1.json 2> $O/prof1.err
cd $R
for d in prof1 prof2; do ls $O/$d; python tools/kernel_breakdown.py $O/$d/r2_kernel_trace.csv 100 warp_fwd_kernel 25 > $O/frame_breakdown_$d.txt 2>&1; cp $O/$d/r2_kernel_stats.csv $O/kernel_stats_$d.csv; rm -f $O/$d/r2_kernel_trace.csv $O/$d/*.db; done
head -3 $O/frame_breakdown_prof1.txt; cut -c1-200 $O/bench_line.json
