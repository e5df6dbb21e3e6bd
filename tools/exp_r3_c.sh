mkdir -p gpurun_out/r3/expc; O=$PWD/gpurun_out/r3/expc; R=$PWD
python -m pytest tests/test_training.py -x -q -m gpu 2>&1 | tail -3
for w in 1 0 1 0; do
  VIDC_TRAIN_WGRAD_STREAM=$w VIDC_TRAIN_PRECISION=bf16 python bench.py --train --batch 8 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bf16 wgrad side stream $w:', d['ms_per_step'], d['value'], d['losses'][-2:])"
done
for w in 1 0; do
  VIDC_TRAIN_WGRAD_STREAM=$w VIDC_TRAIN_PRECISION=fp32 python bench.py --train --batch 8 --steps 5 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 wgrad side stream $w:', d['ms_per_step'], d['value'], d['losses'][-2:])"
done
cd /tmp; export TMPDIR=/tmp
for remap in "" "25:33"; do
  tag=$(echo "x$remap" | tr ':' '_')
  VIDC_TILE_REMAP=$remap VIDC_EXEC=eager rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_$tag -o f --output-format csv -- python3 $R/tools/frame_replay.py 20 > $O/pmc_$tag.log 2>&1
  python3 - <<PY
import csv, collections
per=collections.defaultdict(lambda: [0.0,0.0,0.0,0])
for r in csv.DictReader(open("$O/pmc_$tag/f_counter_collection.csv")):
    n=r["Kernel_Name"]
    if "128, 128" not in n: continue
    k=n.split("(")[0].replace("(anonymous namespace)::","").replace("void ","")
    c=r["Counter_Name"]; v=float(r["Counter_Value"])
    if c=="SQ_VALU_MFMA_BUSY_CYCLES": per[k][0]+=v
    if c=="GRBM_GUI_ACTIVE": per[k][1]+=v; per[k][2]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3; per[k][3]+=1
for k,(b,g,us,n) in per.items():
    print("remap '$remap':", k, "launches", n, "avg us %.1f" % (us/max(n,1)), "MFMA busy %.1f %%" % (100*b/(g/8*1024)))
PY
done
