# round 4, after the last host-side changes (one scheduler, prepare_interleaved, native permutation): the bench LINES again, same box for all.
# (kernel traces / counter tables of evidence_r4.sh are unaffected: no kernel changed.)   bash tools/evidence_r4_lines.sh
set -x
R=$PWD; O=$R/gpurun_out/r4/lines; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_line.json                      # the driver's command
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_line_again.json                # ... twice: run-to-run spread on one box
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 > $O/bench_line_200.json
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs --lanes 1 --per-op $O/per_op.tsv 2>/dev/null | tail -1 > $O/bench_line_200_lanes1.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --frames-per-launch 1 2>/dev/null | tail -1 > $O/bench_line_one_item_per_launch.json
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs --no-sequential-leg --frames-per-launch 4 --lanes 2 2>/dev/null | tail -1 > $O/bench_line_four_items_per_launch_200.json
python bench.py --batch 8 --source 640x480 --height 240 --plane-head --steps 100 --warmup 10 --frames-per-launch 1 --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>/dev/null | tail -1 > $O/bench_line_b8_640x480_plane_head.json
python bench.py --batch 4 --source 1280x720 --height 240 --steps 100 --warmup 10 --frames-per-launch 1 --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>/dev/null | tail -1 > $O/bench_line_b4_1280x720.json
VIDC_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 > $O/bench_line_2ranks_gloo_1gpu.json
VIDC_PRECISION=mixed python tools/host_profile.py 200 2 1 0 2 > $O/host_profile_mixed.txt 2>&1
VIDC_PRECISION=fp32 python tools/group_timeline.py 20 3 2 > $O/timeline_fp32_F2_L3.txt 2>&1
VIDC_PRECISION=mixed python tools/group_timeline.py 20 2 2 > $O/timeline_mixed_F2_L2.txt 2>&1
for f in $O/bench_line*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d['value'], d['dtype'], (d.get('conv_stack') or {}).get('at_measured_frame_rate',{}).get('frac_of_peak_executed'), d.get('value_mixed'), d.get('first_item_latency_ms'))
PY
done
