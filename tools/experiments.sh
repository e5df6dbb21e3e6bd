#!/bin/bash
# Every one-off A/B and tuning run of rounds 3 and 4, one function per experiment (formerly tools/exp_r3_*.sh / exp_r4_*.sh).  Run ON the GPU box
# from the repo root:   bash tools/experiments.sh <name> [args]      bash tools/experiments.sh list
# Results the DESIGN quotes are the files under profiles/ named in its experiment tables (4.4, 4.5, 7.4).  (Function bodies are not
# indented: several contain here-documents.)

r3_b() {
mkdir -p gpurun_out/r3/expb
python -m pytest tests/test_hip_parity.py -x -q -k "lanes_are_bit_identical or first_and_drain or interleaved_matches" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3/expb/bench20.json
python -c "import json; d=json.load(open('gpurun_out/r3/expb/bench20.json')); print('20 steps:', d['value'], d['value_fp32'], d['sequential_call_cnn'], d['fp32_leg']['sequential_call_cnn'])"
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('200 steps:', d['value'], d['value_fp32'])"
R2="6:28,10:28,4:29,7:30,11:30,21:32,22:32"
for remap in "25:1,24:1" "25:1,24:1,$R2" "25:1,24:1,2:29,20:29,17:29,19:31,3:31,$R2"; do
  for lanes in 2 3; do
    VIDC_TILE_REMAP=$remap python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sequential-leg --no-fp32-leg --lanes $lanes 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mixed remap=$remap lanes $lanes:', d['value'], d['program_ms'])"
  done
done
VIDC_TRAIN_PRECISION=bf16 python bench.py --train --batch 8 --steps 5 --warmup 3 2>/dev/null | tail -1 | cut -c1-600
python tools/dump_config2_detections.py gpurun_out/r3/config2_det.npz 2>&1 | tail -1
}

r3_c() {
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
}

r3_d2() {
# round 3 experiment: 2-deep-ring tiles (32-48 KB of LDS, >= 3 workgroups per CU) swapped in for the table's choices, one to three lanes
python -m pytest tests/test_hip_parity.py -q -x -k "conv_tiles and (28- or 29- or 30- or 31- or 32-)" 2>&1 | tail -2
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sequential-leg --no-fp32-leg --lanes $1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VIDC_PRECISION remap=$VIDC_TILE_REMAP lanes $1:', d['value'], d['program_ms'])"; }
R1="6:28,10:28"
R2="6:28,10:28,4:29,7:30,11:30,21:32,22:32"
R3="6:28,10:28,4:29,7:30,11:30,21:32,22:32,3:31"
for prec in fp32 mixed; do
  export VIDC_PRECISION=$prec
  for remap in "" "$R1" "$R2" "$R3"; do
    export VIDC_TILE_REMAP=$remap
    for lanes in 1 2 3; do run $lanes; done
  done
done
}

r3_early() {
# round 3: at the start of a stream a lane takes its second frame as soon as its first is enriched (before the next lane's hypothesis draws)
python -m pytest tests/test_hip_parity.py tests/test_configs.py tests/test_dorn.py -q -x -k "interleaved or lanes or config or golden or variants" 2>&1 | tail -3
run() { python bench.py --steps $1 --warmup 5 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('early=$VIDC_EARLY_SECOND_FRAME steps $1:', d['value'], d['value_fp32'])"; }
for rep in 1 2 3 4; do for e in 0 1; do export VIDC_EARLY_SECOND_FRAME=$e; run 20; done; done
for e in 0 1; do export VIDC_EARLY_SECOND_FRAME=$e; run 200; done
}

r3_lds_cap() {
mkdir -p gpurun_out/r3/exp1
export VIDC_PRECISION=fp32
for lanes in 2 3; do
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sequential-leg --lanes $lanes 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cap none lanes $lanes', d['value'], d['program_ms'])"
  for cap in 80 64; do
    VIDC_LDS_CAP_KB=$cap python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sequential-leg --lanes $lanes 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cap $cap lanes $lanes', d['value'], d['program_ms'])"
  done
done
}

r3_pipelined() {
python -m pytest tests/test_hip_parity.py -q -x -k "(conv_bf16x3 and (33 or 34 or 35 or 36))" 2>&1 | tail -2
python tools/autotune.py --heights 256 --batches 1 --splitk 1,2 --sigs M5120_N768_K6912,M1280_N768_K6912 --verbose --dry 2>&1 | grep "prec 1"
python tools/autotune.py --heights 240 --batches 8 --splitk 1,2 --sigs M38400_N768_K6912,M9600_N768_K6912,M38400_N256_K2304 --verbose --dry 2>&1 | grep "prec 1"
}

r3_plane_side() {
# round 3: the plane block of a lane on a side stream of the lane (beside segment 1) -- A/B on one box
mkdir -p gpurun_out/r3
python -m pytest tests/test_hip_parity.py tests/test_configs.py -q -x -k "interleaved or lanes or config or golden or dense" 2>&1 | tail -3
run() { python bench.py --steps $2 --warmup 10 --no-cpu-baseline --no-sequential-leg --lanes 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('side=$VIDC_PLANE_SIDE_STREAM steps $2:', d['value'], d['value_fp32'])"; }
for rep in 1 2 3; do
  for side in 0 1; do export VIDC_PLANE_SIDE_STREAM=$side; run 2 200; run 2 20; done
done
for side in 0 1; do export VIDC_PLANE_SIDE_STREAM=$side
python bench.py --batch 8 --source 640x480 --height 240 --plane-head --steps 60 --warmup 10 --no-fp32-leg --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('side=$VIDC_PLANE_SIDE_STREAM configs[2]:', d['value'])"
python bench.py --batch 4 --source 1280x720 --height 240 --steps 60 --warmup 10 --no-fp32-leg --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('side=$VIDC_PLANE_SIDE_STREAM configs[3]:', d['value'])"
done
}

r3_prefetch() {
# round 3: Infinity-Cache weight prefetch (vidc_conv_desc.prefetch, engine links every conv to the next one's weights): A/B on one box
O=gpurun_out/r3/prefetch; mkdir -p $O
python -m pytest tests/test_hip_parity.py -q -x -k "prefetch" 2>&1 | tail -3
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sequential-leg --lanes $1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ahead=$VIDC_PREFETCH_AHEAD cap=$VIDC_PREFETCH_MB lanes $1:', d['value'], d['value_fp32'], d['program_ms'], d['fp32_leg']['program_ms'] if 'fp32_leg' in d else '')"; }
export VIDC_PREFETCH_MB=16
for rep in 1 2; do
for ahead in 0 1 2; do
  export VIDC_PREFETCH_AHEAD=$ahead
  run 1; run 2
done
done
export VIDC_PREFETCH_AHEAD=1
for cap in 6 48; do export VIDC_PREFETCH_MB=$cap; run 1; run 2; done
}

r3_train_add() {
# round 3: training step, bf16: the residual add writes the bf16 operand copy of the block output (no cast launch in the next block's convs)
python -m pytest tests/test_training.py -q -x -m gpu 2>&1 | tail -3
run() { python tools/train_bench.py --batch 8 --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VIDC_TRAIN_PRECISION add_bf16=$VIDC_TRAIN_ADD_BF16 streams=$VIDC_TRAIN_STREAMS:', d['ms_per_step'], 'ms per step', d['losses'][-1])"; }
export VIDC_TRAIN_PRECISION=bf16 VIDC_TRAIN_STREAMS=3
for rep in 1 2 3; do for f in 0 1; do export VIDC_TRAIN_ADD_BF16=$f; run; done; done
}

r3_train_bnadd() {
# round 3: training: the Bottleneck tail relu(bn3(.) + identity) inside bn3's apply pass
python -m pytest tests/test_training.py -q -x -m gpu 2>&1 | tail -3
run() { python tools/train_bench.py --batch 8 --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VIDC_TRAIN_PRECISION bn_add_fused=$VIDC_TRAIN_BN_ADD_FUSED:', d['ms_per_step'], 'ms per step', d['losses'][-1])"; }
export VIDC_TRAIN_PRECISION=bf16
for rep in 1 2 3; do for f in 0 1; do export VIDC_TRAIN_BN_ADD_FUSED=$f; run; done; done
export VIDC_TRAIN_PRECISION=fp32
for f in 0 1; do export VIDC_TRAIN_BN_ADD_FUSED=$f; run; done
}

r3_train_dyt() {
# round 3: training step, bf16: (a) dY^T of every conv written by the BatchNorm backward that produces dY, (b) the weight-gradient GEMM
# writes the OIHW gradient in place -- tests + A/B on one box
mkdir -p gpurun_out/r3
python -m pytest tests/test_training.py -q -x -m gpu 2>&1 | tail -3
run() { python tools/train_bench.py --batch 8 --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VIDC_TRAIN_PRECISION dyt_fused=$VIDC_TRAIN_DYT_FUSED inplace=$VIDC_TRAIN_WGRAD_INPLACE streams=$VIDC_TRAIN_STREAMS:', d['ms_per_step'], 'ms per step', d['losses'][-1])"; }
export VIDC_TRAIN_PRECISION=bf16 VIDC_TRAIN_STREAMS=3
for rep in 1 2; do for f in "0 0" "1 0" "1 1"; do set -- $f; export VIDC_TRAIN_DYT_FUSED=$1 VIDC_TRAIN_WGRAD_INPLACE=$2; run; done; done
export VIDC_TRAIN_STREAMS=1
for f in "0 0" "1 1"; do set -- $f; export VIDC_TRAIN_DYT_FUSED=$1 VIDC_TRAIN_WGRAD_INPLACE=$2; run; done
export VIDC_TRAIN_STREAMS=3 VIDC_TRAIN_PRECISION=fp32
for f in "0 0" "1 1"; do set -- $f; export VIDC_TRAIN_DYT_FUSED=$1 VIDC_TRAIN_WGRAD_INPLACE=$2; run; done
}

r3_train_pack() {
# round 3: training: weight re-packing split -- the pyramids' forward copies on the caller's stream, the rest on the fourth lane beside the pyramids
python -m pytest tests/test_training.py -q -x -m gpu 2>&1 | tail -3
run() { python tools/train_bench.py --batch 8 --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VIDC_TRAIN_PRECISION pack_split=$VIDC_TRAIN_PACK_SPLIT:', d['ms_per_step'], 'ms per step', d['losses'][-1])"; }
export VIDC_TRAIN_PRECISION=bf16
for rep in 1 2 3; do for f in 0 1; do export VIDC_TRAIN_PACK_SPLIT=$f; run; done; done
export VIDC_TRAIN_PRECISION=fp32
for f in 0 1; do export VIDC_TRAIN_PACK_SPLIT=$f; run; done
}

r3_train_retune() {
# round 3: the training step's bf16 launches re-measured against all 39 tilings (the table dates from round 2's 27)
O=gpurun_out/r3/train_retune; mkdir -p $O
run() { VIDC_TRAIN_PRECISION=bf16 python tools/train_bench.py --batch 8 --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1:', d['ms_per_step'], 'ms per step', d['losses'][-1])"; }
run before; run before
cp vi_depth_completion_amd/train_tuning.json $O/train_tuning_before.json
timeout 2400 python tools/autotune_train.py --bf16 --remeasure > $O/autotune_train_bf16.log 2>&1; tail -2 $O/autotune_train_bf16.log
cp vi_depth_completion_amd/train_tuning.json $O/train_tuning_after.json
run after; run after
python -m pytest tests/test_training.py -q -x -m gpu 2>&1 | tail -2
}

r3_train_skip() {
# round 3: training, bf16: the BatchNorm backward behind a stride-1 conv without bias writes dY in its two bf16 forms only
python -m pytest tests/test_training.py -q -x -m gpu 2>&1 | tail -3
run() { python tools/train_bench.py --batch 8 --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VIDC_TRAIN_PRECISION skip_f32_dy=$VIDC_TRAIN_SKIP_F32_DY:', d['ms_per_step'], 'ms per step', d['losses'][-1])"; }
export VIDC_TRAIN_PRECISION=bf16
for rep in 1 2 3; do for f in 0 1; do export VIDC_TRAIN_SKIP_F32_DY=$f; run; done; done
}

r3_train_tail() {
# round 3: training step: weight packing with 64-unit blocks for 1x1 convs, 4-channel max-pool backward, 8-lane chunk reduction, unrolled
# stem weight-gradient loop -- tests, step time and the per-kernel table of one single-stream step
mkdir -p gpurun_out/r3/tail; O=$PWD/gpurun_out/r3/tail; R=$PWD
python -m pytest tests/test_training.py -q -x -m gpu 2>&1 | tail -3
run() { python tools/train_bench.py --batch 8 --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VIDC_TRAIN_PRECISION streams=$VIDC_TRAIN_STREAMS:', d['ms_per_step'], 'ms per step', d['losses'][-1])"; }
export VIDC_TRAIN_PRECISION=bf16
VIDC_TRAIN_STREAMS=3 run; VIDC_TRAIN_STREAMS=3 run; VIDC_TRAIN_STREAMS=1 run
VIDC_TRAIN_PRECISION=fp32 VIDC_TRAIN_STREAMS=3 run
cd /tmp && export TMPDIR=/tmp
VIDC_TRAIN_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 $R/tools/train_bench.py --batch 8 --steps 4 --warmup 4 > $O/train_line_prof.json 2> $O/prof.err
cp $(find $O/prof -name 't_kernel_stats.csv') $O/train_kernel_stats_bf16_1stream.csv; rm -rf $O/prof
head -30 $O/train_kernel_stats_bf16_1stream.csv | cut -c1-150
}

r3_train_tickets() {
# round 3: training: chunk sums of the BatchNorm reductions added by the last workgroup of the partial-sum launch (670 launches fewer per step)
python -m pytest tests/test_training.py -q -x -m gpu 2>&1 | tail -3
run() { python tools/train_bench.py --batch 8 --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VIDC_TRAIN_PRECISION bn_tickets=$VIDC_TRAIN_BN_TICKETS streams=$VIDC_TRAIN_STREAMS:', d['ms_per_step'], 'ms per step', d['losses'][-1])"; }
export VIDC_TRAIN_PRECISION=bf16 VIDC_TRAIN_STREAMS=3
for rep in 1 2 3; do for f in 0 1; do export VIDC_TRAIN_BN_TICKETS=$f; run; done; done
export VIDC_TRAIN_STREAMS=1
for f in 0 1; do export VIDC_TRAIN_BN_TICKETS=$f; run; done
export VIDC_TRAIN_PRECISION=fp32 VIDC_TRAIN_STREAMS=3
for f in 0 1; do export VIDC_TRAIN_BN_TICKETS=$f; run; done
}

r3_train_timeline() {
# round 3: timeline of one captured training step (bf16, batch 8): concurrency, per-queue gaps.  Also: 3 lanes at the driver's 20 steps.
R=$PWD; O=$R/gpurun_out/r3/tl; mkdir -p $O
for lanes in 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sequential-leg --lanes $lanes 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('20 steps lanes $lanes:', d['value'], d['value_fp32'])"; done
for lanes in 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sequential-leg --lanes $lanes 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('20 steps lanes $lanes:', d['value'], d['value_fp32'])"; done
cd /tmp && export TMPDIR=/tmp
for streams in 3 1; do
  VIDC_TRAIN_STREAMS=$streams VIDC_TRAIN_PRECISION=bf16 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$streams -o t -- python3 $R/tools/train_bench.py --batch 8 --steps 4 --warmup 4 > $O/train_line_$streams.json 2> $O/prof_$streams.err
  python $R/tools/train_timeline.py $(find $O/prof_$streams -name 't_kernel_trace.csv') > $O/timeline_streams$streams.txt 2>&1
  rm -rf $O/prof_$streams
  tail -1 $O/train_line_$streams.json | cut -c1-200; cat $O/timeline_streams$streams.txt
done
}

r3_train_xt() {
# round 3: training, bf16: 1x1 convs' wgrad right operand transposed from the bf16 operand copy; the copy made once per activation
python -m pytest tests/test_training.py -q -x -m gpu 2>&1 | tail -3
run() { python tools/train_bench.py --batch 8 --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VIDC_TRAIN_PRECISION xt_bf16=$VIDC_TRAIN_XT_BF16:', d['ms_per_step'], 'ms per step', d['losses'][-1])"; }
export VIDC_TRAIN_PRECISION=bf16
for rep in 1 2 3; do for f in 0 1; do export VIDC_TRAIN_XT_BF16=$f; run; done; done
}

r3_train_xt3() {
# round 3: training, bf16: the 3x3 / strided convs' wgrad right operand gathered from the bf16 operand copy as well
python -m pytest tests/test_training.py -q -x -m gpu 2>&1 | tail -3
run() { python tools/train_bench.py --batch 8 --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VIDC_TRAIN_PRECISION xt_bf16=$VIDC_TRAIN_XT_BF16:', d['ms_per_step'], 'ms per step', d['losses'][-1])"; }
export VIDC_TRAIN_PRECISION=bf16
for rep in 1 2 3; do for f in 0 1; do export VIDC_TRAIN_XT_BF16=$f; run; done; done
}

r3_tune_b2() {
# round 3: batch 2 at 320x256 was never measured (cost-model plan): isolated pass for the missing signatures, then the two-lane tuner
O=gpurun_out/r3/tune_b2; mkdir -p $O
run() { python bench.py --batch 2 --steps 100 --warmup 10 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1:', d['value'], d['value_fp32'], d['program_ms'])"; }
run before
timeout 1500 python tools/autotune.py --heights 256 --batches 2 --only-missing > $O/autotune_isolated.log 2>&1; tail -2 $O/autotune_isolated.log
run isolated
cp vi_depth_completion_amd/conv_tuning.json $O/conv_tuning_isolated.json
timeout 1500 python tools/autotune_lanes.py --height 256 --batch 2 --top 12 --budget-s 900 > $O/autotune_lanes.log 2>&1; tail -3 $O/autotune_lanes.log
run lanes_mixed
cp vi_depth_completion_amd/conv_tuning.json $O/conv_tuning_lanes.json
timeout 1500 python tools/autotune_lanes.py --height 256 --batch 2 --top 10 --budget-s 700 --precision fp32 > $O/autotune_lanes_fp32.log 2>&1; tail -3 $O/autotune_lanes_fp32.log
run lanes_fp32
cp vi_depth_completion_amd/conv_tuning.json $O/conv_tuning_final.json
}

r3_tune_b8() {
# round 3: batch-8 / batch-4 signatures (BASELINE configs[2], [3]: 320x240 after the device-side resize) re-measured with the pipelined
# tiles in the candidate set; bench before and after on the same box
mkdir -p gpurun_out/r3/tune_b8
b8() { python bench.py --batch 8 --source 640x480 --height 240 --steps 100 --warmup 10 --no-fp32-leg --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 b8 640x480:', d['value'], d['program_ms'], d['conv_stack']['at_measured_frame_rate'])"; }
b4() { python bench.py --batch 4 --source 1280x720 --height 240 --steps 100 --warmup 10 --no-fp32-leg --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 b4 1280x720:', d['value'], d['program_ms'], d['conv_stack']['at_measured_frame_rate'])"; }
b8 before; b4 before
python tools/autotune.py --heights 240 --batches 4,8 --splitk 1,2,3,4,8 --frame-only > gpurun_out/r3/tune_b8/autotune.log 2>&1
cp vi_depth_completion_amd/conv_tuning.json gpurun_out/r3/tune_b8/conv_tuning.json
tail -2 gpurun_out/r3/tune_b8/autotune.log
b8 after; b4 after
}

r3_tune_b8_lanes() {
# round 3: two-lane tuner at batch 8 (configs[2]) over the 24 most expensive signatures with every tiling; A/B of the table on the same box
O=gpurun_out/r3/tune_b8l; mkdir -p $O
run() { python bench.py --batch 8 --height 240 --source 640x480 --steps 60 --warmup 8 --no-cpu-baseline --no-sequential-leg --no-fp32-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1:', d['value'], d['program_ms'])"; }
cp vi_depth_completion_amd/conv_tuning.json $O/before.json
run before; run before
timeout 2000 python tools/autotune_lanes.py --height 240 --batch 8 --top 24 --budget-s 1500 > $O/autotune_lanes_b8.log 2>&1; grep -v amdgpu.ids $O/autotune_lanes_b8.log | tail -30
cp vi_depth_completion_amd/conv_tuning.json $O/after.json
run after; run after
cp $O/before.json vi_depth_completion_amd/conv_tuning.json; run before_again
cp $O/after.json vi_depth_completion_amd/conv_tuning.json; run after_again
}

r3_tune_detector() {
# round 3 experiment: the plane-mask detector's conv signatures at batch 8 re-measured against every tiling (incl. the round-3 ones),
# BASELINE configs[2] before and after on the same box
mkdir -p gpurun_out/r3/tune_det
run() { python bench.py --batch 8 --height 240 --source 640x480 --plane-head --steps 40 --warmup 5 --no-cpu-baseline --no-sequential-leg --no-fp32-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1:', d['value'], d['program_ms'])"; }
run before; run before
python tools/plane_mask_bench.py --batches 8 2>&1 | tail -3
cp vi_depth_completion_amd/conv_tuning.json gpurun_out/r3/tune_det/before.json
timeout 2400 python tools/autotune.py --detector --merge --heights 240 --batches 8 > gpurun_out/r3/tune_det/autotune.log 2>&1
tail -3 gpurun_out/r3/tune_det/autotune.log
cp vi_depth_completion_amd/conv_tuning.json gpurun_out/r3/tune_det/conv_tuning.json
run after; run after
python tools/plane_mask_bench.py --batches 8 2>&1 | tail -3
}

r3_tune_fp32() {
# round 3: re-measure the exact-fp32 configuration of the frame program's signatures (320x256, batch 1) with the 2-deep-ring tiles and a
# fine split-K grid, then the fp32 leg of bench.py with the new table
mkdir -p gpurun_out/r3/tune_fp32
python tools/autotune.py --heights 256 --batches 1 --splitk 1,2,3,4,5,6,7,8,10,12,16 --fp32-only --frame-only --verbose > gpurun_out/r3/tune_fp32/autotune.log 2>&1
cp vi_depth_completion_amd/conv_tuning.json gpurun_out/r3/tune_fp32/conv_tuning.json
grep "fp32 " gpurun_out/r3/tune_fp32/autotune.log | grep -v prec | awk '{ if ($3" "$4 != "(was "$7" "$8) print }' | head -80
export VIDC_PRECISION=fp32
for st in "20 5" "200 20"; do set -- $st
  python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new table, $1 steps:', d['value'], d['program_ms'])"
done
}

r3_tune_fp32_again() {
# round 3: one more two-lane tuner pass over the fp32 entries (batch 1, 320x256), A/B of the table on the same box at 20 and 200 steps
O=gpurun_out/r3/tune_fp32c; mkdir -p $O
run() { python bench.py --steps $2 --warmup 5 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 steps $2:', d['value'], d['value_fp32'])"; }
cp vi_depth_completion_amd/conv_tuning.json $O/before.json
timeout 1500 python tools/autotune_lanes.py --height 256 --batch 1 --top 40 --budget-s 1100 --precision fp32 > $O/autotune_lanes_fp32.log 2>&1; grep -v amdgpu.ids $O/autotune_lanes_fp32.log | grep -v "> *\([0-9a-zA-Z]*\) *sk\([0-9]*\) .*\1 *sk\2 " | tail -12
cp vi_depth_completion_amd/conv_tuning.json $O/after.json
for rep in 1 2 3; do
  cp $O/before.json vi_depth_completion_amd/conv_tuning.json; run before 20; run before 200
  cp $O/after.json vi_depth_completion_amd/conv_tuning.json; run after 20; run after 200
done
}

r3_tune_mixed() {
# round 3: the frame program's signatures (320x256, batch 1) re-measured in both arithmetic modes with the 2-deep-ring tiles and a fine
# split-K grid; then bench.py with the new table (kept only if the two-lane stream is faster)
mkdir -p gpurun_out/r3/tune_mixed
cp vi_depth_completion_amd/conv_tuning.json gpurun_out/r3/tune_mixed/conv_tuning_before.json
python tools/autotune.py --heights 256 --batches 1 --splitk 1,2,3,4,5,6,7,8,10,12,16 --frame-only > gpurun_out/r3/tune_mixed/autotune.log 2>&1
cp vi_depth_completion_amd/conv_tuning.json gpurun_out/r3/tune_mixed/conv_tuning.json
tail -3 gpurun_out/r3/tune_mixed/autotune.log
for st in "20 5" "200 20"; do set -- $st
  python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new table, $1 steps:', d['value'], d['value_fp32'], d['program_ms'])"
done
cp gpurun_out/r3/tune_mixed/conv_tuning_before.json vi_depth_completion_amd/conv_tuning.json
for st in "20 5" "200 20"; do set -- $st
  python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old table, $1 steps:', d['value'], d['value_fp32'], d['program_ms'])"
done
}

r3_variants() {
mkdir -p gpurun_out/r3/exp5
python -m pytest tests/test_hip_parity.py -x -q -k "first_and_drain or lanes_are_bit_identical or interleaved_matches or interleaved_golden" 2>&1 | tail -3
for v in 1 0; do
  VIDC_TICK_VARIANTS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variants $v 20 steps:', d['value'], d['value_fp32'])"
done
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variants 1 200 steps:', d['value'], d['value_fp32'])"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sequential-leg --lanes 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variants 1 20 steps lanes 3:', d['value'], d['value_fp32'])"
}

r3_xb() {
# round 3 A/B: the fp32 cross-barrier fragment prefetch (VIDC_FP32_XB=1 build, `make xb`) against the product library on one box
python -m pytest tests/test_hip_parity.py -q -x -k "conv_tiles or conv_splitk or conv_is_deterministic or pipelined_fragment or conv_splitk_shared" 2>&1 | tail -2
python -m pytest tests/test_training.py -q -x -k "plain_bf16_conv_mode" 2>&1 | tail -2
export VIDC_PRECISION=fp32
for lib in libvidc_xb.so libvidc.so libvidc_xb.so libvidc.so; do      # (make -C vi_depth_completion_amd/csrc xb first: libvidc_xb.so = with the prefetch)
  for st in "20 5" "200 20"; do set -- $st
    VIDC_LIB_NAME=$lib python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-sequential-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib fp32 $1 steps:', d['value'], d['program_ms'], 'rmse', d['rmse_vs_oracle'])"
  done
done
}

r4_bnfold() {
python -m pytest tests/test_training.py -x -q -k "folded_batchnorm or training_iteration_vs_reference or plain_bf16_training_mode or graph_replay" 2>&1 | tail -5
for rep in 1 2; do for F in 1 0; do
VIDC_TRAIN_BN_FOLD=$F VIDC_TRAIN_PRECISION=bf16 python bench.py --train --batch 8 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bf16 fold=$F:', d['ms_per_step'], d['value'], d['losses'][-2:])"
done; done
for F in 1 0; do
VIDC_TRAIN_BN_FOLD=$F VIDC_TRAIN_PRECISION=fp32 python bench.py --train --batch 8 --steps 5 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fp32 fold=$F:', d['ms_per_step'], d['value'], d['losses'][-2:])"
done
}

r4_lanes() {
mkdir -p gpurun_out
python -m pytest tests/test_frames_per_launch.py -x -q 2>&1 | tail -3
VIDC_PRECISION=fp32 python tools/group_timeline.py 20 3 2 > gpurun_out/r4_timeline_fp32_F2_L3_b.txt 2>&1
grep "=== rep" gpurun_out/r4_timeline_fp32_F2_L3_b.txt
for L in 2 3 4; do for K in 20 200; do
python bench.py --steps $K --warmup 5 --lanes $L --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lanes $L K $K: fp32', d['value'], d['conv_stack']['at_measured_frame_rate']['frac_of_peak_executed'], ' mixed', d['value_mixed'])"
done; done
}

r4_newtests() {
mkdir -p gpurun_out
python -m pytest tests/test_checkpoints.py tests/test_frames_per_launch.py tests/test_configs.py "tests/test_training.py::test_two_ranks_through_training_steps" "tests/test_training.py::test_two_ranks_with_bf16_gradient_buckets" "tests/test_training.py::test_bf16_gradient_buckets_keep_the_loss_curve" tests/test_dorn.py -x -q -s 2>&1 | tail -40 > gpurun_out/r4_newtests.log
tail -30 gpurun_out/r4_newtests.log
python -m pytest tests/test_hip_parity.py -x -q -k "golden or demo" 2>&1 | tail -5
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err
tail -3 gpurun_out/r4_bench_default.err
python -c "
import json; d=json.loads(open('gpurun_out/r4_bench_default.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','dtype','ms_per_step','rmse_vs_oracle','value_mixed','rmse_vs_oracle_mixed','first_item_latency_ms']})
print(d['conv_stack']['at_measured_frame_rate'], d['config']['lanes'], d['mixed_leg']['lanes'])
print(json.dumps(d.get('extra_legs'), indent=1)[:3000])
print(d['cpu_baseline'])
"
}

r4_pairing() {
# round 4: frames_per_launch (pipeline.run_interleaved) -- tests, then the bench at F = 1, 2, 3, 4 and 20 / 200 steps
mkdir -p gpurun_out
python -m pytest tests/test_frames_per_launch.py "tests/test_configs.py::test_frame_output_does_not_depend_on_the_shard" -x -q 2>&1 | tail -15 > gpurun_out/r4_pairing_tests.log
cat gpurun_out/r4_pairing_tests.log
for F in 2 1 3 4; do
  for K in 20 200; do
    python bench.py --steps $K --warmup 5 --frames-per-launch $F --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>gpurun_out/r4_pairing_F${F}_K${K}.err | tail -1 > gpurun_out/r4_pairing_F${F}_K${K}.json
    python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r4_pairing_F${F}_K${K}.json"))
    print("F=${F} K=${K}: fp32 %.1f fps (frac %.4f, first item %.2f ms, tick %s)  mixed %.1f fps (first item %.2f ms)" % (
        d["value"], d["conv_stack"]["at_measured_frame_rate"]["frac_of_peak_executed"], d["first_item_latency_ms"], d["program_ms"],
        d["value_mixed"], d["mixed_leg"]["first_item_latency_ms"]))
except Exception as e:
    print("F=${F} K=${K}: failed", e)
PY
  done
done 2>&1 | tee gpurun_out/r4_pairing_sweep.txt
}

r4_perop() {
mkdir -p gpurun_out
python -m pytest tests/test_frames_per_launch.py -x -q 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sequential-leg --no-extra-legs --per-op gpurun_out/r4_per_op_F2.tsv 2>gpurun_out/r4_perop.err | tail -1 > gpurun_out/r4_perop_line.json
python -c "
import json; d=json.load(open('gpurun_out/r4_perop_line.json')); print(d['value'], d['value_mixed'], d['conv_stack'], d['roofline'])"
}

r4_stagger() {
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sequential-leg --no-extra-legs --no-mixed-leg > /dev/null 2>&1   # warm the box
for rep in 1 2 3 4; do for S in 0 2 1; do
VIDC_FILL_STAGGER=$S python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sequential-leg --no-extra-legs --no-mixed-leg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('stagger $S: fp32', d['value'], d['conv_stack']['at_measured_frame_rate']['frac_of_peak_executed'], d['first_item_latency_ms'])"
done; done
}

r4_timeline() {
mkdir -p gpurun_out
VIDC_PRECISION=fp32 python tools/group_timeline.py 20 2 2 > gpurun_out/r4_timeline_fp32_F2_L2.txt 2>&1
VIDC_PRECISION=fp32 python tools/group_timeline.py 20 3 2 > gpurun_out/r4_timeline_fp32_F2_L3.txt 2>&1
for L in 3; do for K in 20 200; do
python bench.py --steps $K --warmup 5 --lanes $L --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lanes $L K $K:', d['value'], d['conv_stack']['at_measured_frame_rate']['frac_of_peak_executed'], d['value_mixed'])"
done; done
}

r4_waves3() {
# A/B of a register budget on the 2-deep-ring small conv tiles (csrc/conv_mfma.hip, comment above conv_igemm_f32): build with the
# amdgpu_waves_per_eu(3) line in place, then:
python -m pytest tests/test_hip_parity.py -x -q -k "conv_tiles or conv_splitk or conv_is_deterministic or shared_workspace" 2>&1 | tail -3
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K20: fp32', d['value'], d['program_ms'], ' mixed', d['value_mixed'], d['mixed_leg']['program_ms'])"
done
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-sequential-leg --no-extra-legs --per-op gpurun_out/r4_per_op_w3.tsv 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K200: fp32', d['value'], ' mixed', d['value_mixed'])"
}

r4_schedulers_f1() {
# round 4: the round-3 scheduler (VIDC_GROUPED_SCHEDULER=0, removed after this run: git history) against the round-4 one at one item per
# launch -- profiles/r4_schedulers_one_item_per_launch.txt
for rep in 1 2; do for G in 0 1; do for L in 2 3; do
VIDC_GROUPED_SCHEDULER=$G python bench.py --steps 20 --warmup 5 --lanes $L --frames-per-launch 1 --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('F=1 grouped=$G lanes $L K20: fp32', d['value'], ' mixed', d['value_mixed'])"
done; done; done
for G in 0 1; do
VIDC_GROUPED_SCHEDULER=$G python bench.py --batch 8 --source 640x480 --height 240 --plane-head --steps 40 --warmup 6 --frames-per-launch 1 --lanes 2 --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('configs[2] grouped=$G: fp32', d['value'], ' mixed', d['value_mixed'])"
VIDC_GROUPED_SCHEDULER=$G python bench.py --batch 4 --source 1280x720 --height 240 --steps 60 --warmup 6 --frames-per-launch 1 --lanes 2 --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('configs[3] grouped=$G: fp32', d['value'], ' mixed', d['value_mixed'])"
done
}

r4_four_items() {
# round 4: table entries for M = 1280 (isolated tuner), then four items per launch at 20 / 200 / 400 steps -- profiles/r4_four_items_per_launch.txt
mkdir -p gpurun_out
timeout 2400 python tools/autotune.py --heights 256 --batches 4 --only-missing --frame-only --out gpurun_out/conv_tuning_f4.json > gpurun_out/r4_autotune_b4_256.log 2>&1
tail -3 gpurun_out/r4_autotune_b4_256.log
cp gpurun_out/conv_tuning_f4.json vi_depth_completion_amd/conv_tuning.json
for L in 2 3; do for K in 20 200 400; do
python bench.py --steps $K --warmup 8 --lanes $L --frames-per-launch 4 --no-cpu-baseline --no-sequential-leg --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('F=4 lanes $L K $K: fp32', d['value'], d['conv_stack']['at_measured_frame_rate']['frac_of_peak_executed'], d['program_ms'], d['first_item_latency_ms'], ' mixed', d['value_mixed'], d['mixed_leg']['program_ms'])"
done; done
}

r4_hw_queues() {
# round 4: GPU_MAX_HW_QUEUES against the inference stream mode and the training step (one graph / two graphs through a one-rank RCCL group)
# -- profiles/r4_lane_streams_hw_queues.txt, r4_train_side_stream_experiments.txt
line() { grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d.get('value'), d.get('value_mixed'), d.get('ms_per_step'))"; }
for Q in 2 3 4 5 6 8; do GPU_MAX_HW_QUEUES=$Q python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-sequential-leg 2>/dev/null | line "inference Q=$Q:"; done
for Q in 3 4 5 6 8 12; do
GPU_MAX_HW_QUEUES=$Q VIDC_TRAIN_PRECISION=bf16 python bench.py --train --batch 8 --steps 10 --warmup 3 2>/dev/null | line "train, one graph Q=$Q:"
GPU_MAX_HW_QUEUES=$Q VIDC_DIST_WORLD1=1 VIDC_TRAIN_PRECISION=bf16 python bench.py --train --batch 8 --steps 10 --warmup 3 2>/dev/null | line "train, two graphs + RCCL world 1 Q=$Q:"
done
}

r4_rccl_group() {
# round 4: the stream mode with and without a (one-rank) RCCL process group in the process -- profiles/r4_lane_streams_hw_queues.txt
for rep in 1 2; do for w in 0 1; do
VIDC_DIST_WORLD1=$w python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-sequential-leg 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('RCCL world-1 group $w:', d['value'], d['value_mixed'])"
done; done
}

r4_items_and_lanes() {
# round 4: items per launch x lanes at 20 / 200 steps (DESIGN 5.1's table)
for K in 20 200; do for FL in "2 3" "4 2" "4 3"; do set -- $FL
python bench.py --gpus 1 --steps $K --warmup $((K/10+3)) --no-cpu-baseline --no-extra-legs --no-sequential-leg --frames-per-launch $1 --lanes $2 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('F=$1 lanes=$2 K=$K:', d['value'], d['value_mixed'], d['first_item_latency_ms'])"
done; done
}

case "$1" in
  list|"") echo "experiments: r4_hw_queues r4_rccl_group r4_items_and_lanes r4_four_items r4_schedulers_f1 r4_waves3 r3_b r3_c r3_d2 r3_early r3_lds_cap r3_pipelined r3_plane_side r3_prefetch r3_train_add r3_train_bnadd r3_train_dyt r3_train_pack r3_train_retune r3_train_skip r3_train_tail r3_train_tickets r3_train_timeline r3_train_xt r3_train_xt3 r3_tune_b2 r3_tune_b8 r3_tune_b8_lanes r3_tune_detector r3_tune_fp32 r3_tune_fp32_again r3_tune_mixed r3_variants r3_xb r4_bnfold r4_lanes r4_newtests r4_pairing r4_perop r4_stagger r4_timeline" ;;
  *) name="$1"; shift; if declare -F "$name" > /dev/null; then "$name" "$@"; else echo "unknown experiment $name (bash tools/experiments.sh list)"; exit 2; fi ;;
esac
