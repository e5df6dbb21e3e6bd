#!/bin/bash
# A/B on ONE box: the F(4x4) layers of the small maps as three launches (table) against ONE launch (csrc/wfused.hip, VIDC_WINO_FUSED = largest tile count
# that is fused), in the stream mode and on one lane.
cd "$(dirname "$0")/.."
O=gpurun_out/ab_wfused; mkdir -p $O
run() {
  tag=$1; knob=$2
  for rep in 1 2; do VIDC_WINO_FUSED=$knob python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-sequential-leg --no-mixed-leg 2>$O/err_${tag}_$rep.txt | grep '^{' | tail -1 > $O/line_${tag}_$rep.json; done
  VIDC_WINO_FUSED=$knob python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extra-legs --no-sequential-leg --no-mixed-leg --lanes 1 --per-op $O/per_op_$tag.tsv 2>$O/err_${tag}_l1.txt | grep '^{' | tail -1 > $O/line_${tag}_lanes1.json
}
run base 0
run f80 80
run f320 320
run base2 0
python - <<'PY' > gpurun_out/ab_wfused/summary.txt 2>&1
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_wfused/line_*.json")):
    try:
        d=json.loads(open(f).read()); print(f.split('/')[-1], d.get("value"), d.get("steady_state_frames_per_s"), d["roofline"].get("avg_launch_us"), d.get("rmse_vs_oracle"), d.get("program_ms"))
    except Exception as e: print(f, "FAILED", e)
def load(p): return [(x.split('\t')[2], float(x.split('\t')[1])) for x in open(p).read().splitlines()]
A=load("gpurun_out/ab_wfused/per_op_base.tsv.fp32")
for tag in ("f80","f320","base2"):
    try:
        B=load("gpurun_out/ab_wfused/per_op_%s.tsv.fp32" % tag)
        print("one-lane tick us: base %.1f (%d ops)  %s %.1f (%d ops)" % (sum(t for _,t in A), len(A), tag, sum(t for _,t in B), len(B)))
    except Exception as e: print(tag, "FAILED", e)
PY
cat gpurun_out/ab_wfused/summary.txt; tail -3 $O/err_f80_1.txt
