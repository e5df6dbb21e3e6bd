#!/bin/bash
# A/B on ONE box: the fp32 conv loop with its DMA instructions in MFMA groups 0 and 1 (default) against spread over all four (-DVIDC_CONV_SPREAD_DMA)
cd "$(dirname "$0")/.."
O=gpurun_out/ab_spread; mkdir -p $O
run() {
  tag=$1
  for rep in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-sequential-leg --no-mixed-leg 2>/dev/null | grep '^{' | tail -1 > $O/line_${tag}_$rep.json; done
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extra-legs --no-sequential-leg --no-mixed-leg --lanes 1 --per-op $O/per_op_$tag.tsv 2>/dev/null | grep '^{' | tail -1 > $O/line_${tag}_lanes1.json
}
run base
( cd vi_depth_completion_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DVIDC_CONV_SPREAD_DMA -c conv_mfma.hip -o conv_mfma.o > /dev/null 2>&1 && make > /dev/null 2>&1 )
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "conv_tiles or splitk" 2>&1 | tail -2
run spread
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_spread/line_*.json")):
    try:
        d=json.loads(open(f).read()); print(f.split('/')[-1], d.get("value"), d.get("steady_state_frames_per_s"), d["roofline"].get("avg_launch_us"))
    except Exception as e: print(f, "FAILED", e)
def load(p): return [(x.split('\t')[2], float(x.split('\t')[1])) for x in open(p).read().splitlines()]
A=load("gpurun_out/ab_spread/per_op_base.tsv.fp32"); B=load("gpurun_out/ab_spread/per_op_spread.tsv.fp32")
print("one-lane tick us: base %.1f spread %.1f" % (sum(t for _,t in A), sum(t for _,t in B)))
d=sorted(((tb-ta,n,ta,tb) for (n,ta),(_,tb) in zip(A,B)))
for x in d[:8]+d[-8:]: print("%+7.2f %-90s %7.2f -> %7.2f" % (x[0], x[1][:90], x[2], x[3]))
PY
