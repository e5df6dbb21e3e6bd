# round 3: the fp32 (reference-arithmetic) leg under the profiler -- per-op table, kernel trace with stats (one and two lanes)
set -x
R=$PWD; O=$R/gpurun_out/r3/fp32; mkdir -p $O
export VIDC_PRECISION=fp32
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --lanes 1 --per-op $O/per_op 2>/dev/null | tail -1 > $O/bench_line_fp32_lanes1.json
python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_fp32_200.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_fp32_20.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o r3 -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --lanes 1 > $O/bench_line_fp32_profiled_lanes1.json 2> $O/prof1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -o r3 -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_line_fp32_profiled.json 2> $O/prof2.err
cd $R
for d in prof1 prof2; do ls $O/$d; python tools/kernel_breakdown.py $O/$d/r3_kernel_trace.csv 100 warp_fwd_kernel 25 > $O/frame_breakdown_$d.txt 2>&1; cp $O/$d/r3_kernel_stats.csv $O/kernel_stats_$d.csv; rm -f $O/$d/r3_kernel_trace.csv $O/$d/*.db; done
head -5 $O/frame_breakdown_prof1.txt; cut -c1-300 $O/bench_line_fp32_20.json
