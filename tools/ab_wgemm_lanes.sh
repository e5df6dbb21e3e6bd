#!/bin/bash
# A/B on ONE box: the few-row Winograd products (M = 80) of the frame program on the general tile (table) against the streamed tiles of csrc/wgemm.hip
# (VIDC_TUNING_OVERRIDE), in the stream mode -- where what counts is CU time beside two other lanes, not the latency of a launch alone.
cd "$(dirname "$0")/.."
O=gpurun_out/ab_wgemm; mkdir -p $O
python tools/wgemm_bench.py --tiles 28,40,41 --chunks 32,48,64,96,128 --iters 60 > $O/wgemm_bench.txt 2>&1
OV40='{"M80_N256_K256_k1s1_G144":[40,64],"M80_N512_K512_k1s1_G64":[40,64],"M80_N512_K512_k1s1_G72":[40,64],"M80_N1024_K1024_k1s1_G16":[40,64],"M80_N3072_K3072_k1s1_G16":[40,64],"M80_N1536_K1536_k1s1_G72":[40,64]}'
OV41=${OV40//\[40,/[41,}
OVL3='{"M80_N256_K256_k1s1_G144":[40,64]}'
run() {
  tag=$1; ov=$2
  for rep in 1 2; do VIDC_TUNING_OVERRIDE="$ov" python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-sequential-leg --no-mixed-leg 2>$O/err_${tag}_$rep.txt | grep '^{' | tail -1 > $O/line_${tag}_$rep.json; done
  VIDC_TUNING_OVERRIDE="$ov" python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extra-legs --no-sequential-leg --no-mixed-leg --lanes 1 --per-op $O/per_op_$tag.tsv 2>/dev/null | grep '^{' | tail -1 > $O/line_${tag}_lanes1.json
}
run base '{}'
run s40 "$OV40"
run s41 "$OV41"
run l3s40 "$OVL3"
run base2 '{}'
python - <<'PY' > gpurun_out/ab_wgemm/summary.txt 2>&1
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_wgemm/line_*.json")):
    try:
        d=json.loads(open(f).read()); print(f.split('/')[-1], d.get("value"), d.get("steady_state_frames_per_s"), d["roofline"].get("avg_launch_us"), d.get("rmse_vs_oracle"))
    except Exception as e: print(f, "FAILED", e)
def load(p): return [(x.split('\t')[2], float(x.split('\t')[1])) for x in open(p).read().splitlines()]
A=load("gpurun_out/ab_wgemm/per_op_base.tsv.fp32")
for tag in ("s40","s41","l3s40","base2"):
    try:
        B=load("gpurun_out/ab_wgemm/per_op_%s.tsv.fp32" % tag)
        print("one-lane tick us: base %.1f %s %.1f" % (sum(t for _,t in A), tag, sum(t for _,t in B)))
        d=sorted(((tb-ta,n,ta,tb) for (n,ta),(_,tb) in zip(A,B)))
        for x in d[:5]+d[-5:]: print("   %+7.2f %-100s %7.2f -> %7.2f" % (x[0], x[1][:100], x[2], x[3]))
    except Exception as e: print(tag, "FAILED", e)
PY
cat gpurun_out/ab_wgemm/summary.txt; cat $O/wgemm_bench.txt | cut -c1-400
