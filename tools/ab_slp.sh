#!/bin/bash
# A/B on ONE box: the library as built (-fno-slp-vectorize) against the same sources with hipcc's SLP vectoriser on; per-op tables of the fp32 tick
cd "$(dirname "$0")/.."
O=gpurun_out/ab_slp; mkdir -p $O
run() {
  tag=$1
  for rep in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-sequential-leg 2>/dev/null | grep '^{' | tail -1 > $O/line_${tag}_$rep.json; done
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extra-legs --no-sequential-leg --no-mixed-leg --lanes 1 --per-op $O/per_op_$tag.tsv 2>/dev/null | grep '^{' | tail -1 > $O/line_${tag}_lanes1.json
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-sequential-leg --no-mixed-leg --frames-per-launch 1 2>/dev/null | grep '^{' | tail -1 > $O/line_${tag}_F1.json
}
run noslp
sed -i 's/ -fno-slp-vectorize//' vi_depth_completion_amd/csrc/Makefile
make -C vi_depth_completion_amd/csrc clean > /dev/null; make -C vi_depth_completion_amd/csrc -j32 > $O/make_slp.log 2>&1
run slp
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_slp/line_*.json")):
    try:
        d=json.loads(open(f).read()); print(f.split('/')[-1], d.get("value"), d.get("steady_state_frames_per_s"), d.get("value_mixed"))
    except Exception as e: print(f, "FAILED", e)
PY
