#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/stale
R=tools/stale_read/repro
M=gpurun_out/stale/matrix9.txt
: > $M
BASE="--lanes 3 --F 4 --items 1200 --nf0 4 --nf1 2 --pattern raw"
for gap in 0 1; do
  for nz in -1 1 2 5 9 13 14; do
    timeout 300 $R $BASE --gap $gap --noise $nz --verbose 2>&1 | grep -E "RESULT|words from|want" | head -8 | sed "s/^RESULT/RESULT mode=$gap/" >> $M || echo "   (exit $?)" >> $M
  done
done
cat > /tmp/conv_noise.py <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from vi_depth_completion_amd import ops
prec = int(sys.argv[1]); secs = float(sys.argv[2])
x = torch.randn(4, 60, 80, 256, device="cuda"); w = torch.randn(256, 256, 3, 3, device="cuda") * 0.02
one, zero = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
wp = ops.pack_conv_weight_bf16x3(w) if prec else ops.pack_conv_weight(w)
t0 = time.time(); n = 0
print("conv noise prec", prec, "running", flush=True)
while time.time() - t0 < secs:
    for _ in range(50):
        ops.conv2d_bn_act(x, wp, one, zero, 3, 3, pad=1, relu1=True, precision=prec)
    torch.cuda.synchronize(); n += 50
print("conv noise prec", prec, "launches", n, flush=True)
PY
python /tmp/conv_noise.py 1 30 >> $M 2>&1 &
NP=$!
sleep 12
for gap in 0 1; do
  timeout 300 $R --lanes 3 --F 4 --items 2400 --nf0 4 --nf1 2 --pattern raw --gap $gap --verbose 2>&1 | grep -E "RESULT|words from|want" | head -8 | sed "s/^RESULT/RESULT conv_process mode=$gap/" >> $M
done
wait $NP
cat $M | cut -c1-250
