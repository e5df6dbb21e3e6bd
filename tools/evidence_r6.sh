# round 6, four stream items per launch (bench.py's default since): tests of the stream mode, kernel traces, in-frame counters at program
# batch 4, the per-op table and the bench lines, one box.   bash tools/evidence_r6_f4.sh
R=$PWD; O=$R/gpurun_out/r6/f4; mkdir -p $O
python -m pytest tests/test_frames_per_launch.py tests/test_configs.py -x -q -m gpu > $O/pytest_stream_mode.log 2>&1; tail -2 $O/pytest_stream_mode.log
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 > $O/bench_line.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 > $O/bench_line_second_run.json
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs 2>/dev/null | grep '^{' | tail -1 > $O/bench_line_200.json
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs --lanes 1 --per-op $O/per_op.tsv 2>/dev/null | grep '^{' | tail -1 > $O/bench_line_200_lanes1.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --frames-per-launch 2 2>/dev/null | grep '^{' | tail -1 > $O/bench_line_two_items_per_launch.json
VIDC_DIST_WORLD1=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | grep '^{' | tail -1 > $O/bench_line_rccl_world1.json
VIDC_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-sequential-leg 2>/dev/null | grep '^{' | tail -1 > $O/bench_line_2ranks_gloo_1gpu.json
VIDC_PRECISION=fp32 python tools/group_timeline.py 20 3 4 > $O/timeline_fp32_F4_L3.txt 2>&1
cd /tmp && export TMPDIR=/tmp
VIDC_PRECISION=fp32 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fp32 -o r6 -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-sequential-leg --no-extra-legs --lanes 1 > $O/bench_line_fp32_profiled_lanes1.json 2> $O/prof_fp32.err
VIDC_PRECISION=mixed rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mixed -o r6 -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-sequential-leg --no-extra-legs --lanes 1 > $O/bench_line_mixed_profiled_lanes1.json 2> $O/prof_mixed.err
VIDC_PRECISION=fp32 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fp32_3 -o r6 -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-sequential-leg --no-extra-legs > $O/bench_line_fp32_profiled.json 2> $O/prof_fp32_3.err
cd $R
for d in prof_fp32 prof_mixed prof_fp32_3; do python tools/kernel_breakdown.py $(find $O/$d -name 'r6_kernel_trace.csv') 25 warp_fwd_kernel 12 > $O/frame_breakdown_$d.txt 2>&1; cp $(find $O/$d -name 'r6_kernel_stats.csv') $O/kernel_stats_$d.csv; rm -rf $O/$d; done
cd /tmp
for mode in fp32 mixed; do
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $pass | cut -d' ' -f1)
    VIDC_PRECISION=$mode VIDC_EXEC=eager rocprofv3 --kernel-trace --pmc $pass -d $O/pmc_${mode}_$tag -o f --output-format csv -- python3 $R/tools/frame_replay.py 20 > $O/pmc_${mode}_$tag.log 2>&1
  done
  python $R/tools/frame_pmc_summary.py $(find $O/pmc_${mode}_FETCH_SIZE -name 'f_counter_collection.csv') $(find $O/pmc_${mode}_WRITE_SIZE -name 'f_counter_collection.csv') $(find $O/pmc_${mode}_SQ_VALU_MFMA_BUSY_CYCLES -name 'f_counter_collection.csv') 20 > $O/frame_pmc_$mode.txt 2>&1
  rm -rf $O/pmc_${mode}_FETCH_SIZE $O/pmc_${mode}_WRITE_SIZE $O/pmc_${mode}_SQ_VALU_MFMA_BUSY_CYCLES
done
cd $R
CMD="rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE> -- python3 tools/frame_replay.py 20 (VIDC_EXEC=eager, program batch 4"
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python tools/pmc_to_json.py $O/frame_pmc_fp32.txt fp32 "$CMD, VIDC_PRECISION=fp32; profiles/r6_frame_pmc_fp32.txt)" --out $O/pmc_traffic.json
python tools/pmc_to_json.py $O/frame_pmc_mixed.txt mixed "$CMD, VIDC_PRECISION=mixed; profiles/r6_frame_pmc_mixed.txt)" --out $O/pmc_traffic.json
head -5 $O/frame_pmc_fp32.txt $O/frame_pmc_mixed.txt
for f in $O/bench_line*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['dtype'], (d.get('conv_stack') or {}).get('at_measured_frame_rate',{}).get('frac_of_peak_executed'), d.get('value_mixed'), d.get('first_item_latency_ms'), d['roofline'].get('kernel'), d['roofline'].get('frac'), d['roofline'].get('traffic'))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
# the bandwidth-bound kernels incl. the Winograd transforms: HIP-event rates and rocprofv3 FETCH / WRITE passes
python tools/glue_bench.py > $O/glue_hbm.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/A -o g --output-format csv -- python3 $R/tools/glue_bench.py > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/B -o g --output-format csv -- python3 $R/tools/glue_bench.py > $O/b.log 2>&1
cd $R
python tools/glue_pmc_summary.py $(find $O/A -name 'g_counter_collection.csv') $(find $O/B -name 'g_counter_collection.csv') > $O/glue_pmc.txt 2>&1
rm -rf $O/A $O/B
head -60 $O/glue_pmc.txt
