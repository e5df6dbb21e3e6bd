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
