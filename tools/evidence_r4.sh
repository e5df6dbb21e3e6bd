# round 4 evidence: every number DESIGN.md / profiles/README.md quote, from one GPU box.  Usage (on the GPU box): bash tools/evidence_r4.sh
set -x
R=$PWD; O=$R/gpurun_out/r4/final; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_line.json                      # the driver's command
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 > $O/bench_line_200.json
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs --lanes 1 --per-op $O/per_op.tsv 2>/dev/null | tail -1 > $O/bench_line_200_lanes1.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --frames-per-launch 1 2>/dev/null | tail -1 > $O/bench_line_one_item_per_launch.json
python bench.py --batch 8 --source 640x480 --height 240 --plane-head --steps 100 --warmup 10 --frames-per-launch 1 --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>/dev/null | tail -1 > $O/bench_line_b8_640x480_plane_head.json
python bench.py --batch 4 --source 1280x720 --height 240 --steps 100 --warmup 10 --frames-per-launch 1 --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>/dev/null | tail -1 > $O/bench_line_b4_1280x720.json
VIDC_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 > $O/bench_line_2ranks_gloo_1gpu.json
VIDC_TRAIN_PRECISION=bf16 python bench.py --train --batch 8 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/train_line_bf16_b8.json
VIDC_TRAIN_PRECISION=fp32 python bench.py --train --batch 8 --steps 5 --warmup 3 2>/dev/null | tail -1 > $O/train_line_fp32_b8.json
python tools/group_timeline.py 20 3 2 > $O/timeline_mixed_F2_L3.txt 2>&1
VIDC_PRECISION=fp32 python tools/group_timeline.py 20 3 2 > $O/timeline_fp32_F2_L3.txt 2>&1
cd /tmp && export TMPDIR=/tmp
# kernel traces (one lane: per-kernel durations that are not stretched by the other lanes' launches)
VIDC_PRECISION=fp32 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fp32 -o r4 -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-sequential-leg --no-extra-legs --lanes 1 > $O/bench_line_fp32_profiled_lanes1.json 2> $O/prof_fp32.err
VIDC_PRECISION=mixed rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mixed -o r4 -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-sequential-leg --no-extra-legs --lanes 1 > $O/bench_line_mixed_profiled_lanes1.json 2> $O/prof_mixed.err
VIDC_PRECISION=fp32 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fp32_3 -o r4 -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-sequential-leg --no-extra-legs > $O/bench_line_fp32_profiled.json 2> $O/prof_fp32_3.err
cd $R
for d in prof_fp32 prof_mixed prof_fp32_3; do python tools/kernel_breakdown.py $O/$d/r4_kernel_trace.csv 50 warp_fwd_kernel 25 > $O/frame_breakdown_$d.txt 2>&1; cp $O/$d/r4_kernel_stats.csv $O/kernel_stats_$d.csv; rm -rf $O/$d; done
# counters in the frame, one group per pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains besides --kernel-trace)
cd /tmp
for mode in fp32 mixed; do
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $pass | cut -d' ' -f1)
    VIDC_PRECISION=$mode VIDC_EXEC=eager rocprofv3 --kernel-trace --pmc $pass -d $O/pmc_${mode}_$tag -o f --output-format csv -- python3 $R/tools/frame_replay.py 20 > $O/pmc_${mode}_$tag.log 2>&1
  done
  python $R/tools/frame_pmc_summary.py $O/pmc_${mode}_FETCH_SIZE/f_counter_collection.csv $O/pmc_${mode}_WRITE_SIZE/f_counter_collection.csv $O/pmc_${mode}_SQ_VALU_MFMA_BUSY_CYCLES/f_counter_collection.csv > $O/frame_pmc_$mode.txt 2>&1
  rm -rf $O/pmc_${mode}_FETCH_SIZE $O/pmc_${mode}_WRITE_SIZE $O/pmc_${mode}_SQ_VALU_MFMA_BUSY_CYCLES
done
cd $R
CMD="rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE> -- python3 tools/frame_replay.py 20 (VIDC_EXEC=eager, program batch 2"
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python tools/pmc_to_json.py $O/frame_pmc_fp32.txt fp32 "$CMD, VIDC_PRECISION=fp32; profiles/r4_frame_pmc_fp32.txt)" --out $O/pmc_traffic.json
python tools/pmc_to_json.py $O/frame_pmc_mixed.txt mixed "$CMD, VIDC_PRECISION=mixed; profiles/r4_frame_pmc_mixed.txt)" --out $O/pmc_traffic.json
python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu.log
head -4 $O/frame_pmc_fp32.txt $O/frame_pmc_mixed.txt; cut -c1-200 $O/bench_line.json; cat $O/pytest_gpu.log
