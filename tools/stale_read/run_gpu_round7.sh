#!/bin/bash
# the WAR pattern (address VGPRs of a global_load overwritten by the next VALU) beside synthetic noise, then beside the real bf16x3 conv kernel (another process)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/stale
R=tools/stale_read/repro
M=gpurun_out/stale/matrix7.txt
: > $M
BASE="--lanes 3 --F 4 --items 1200 --nf0 4 --nf1 2 --pattern war"
for gap in 0 1 2 4; do
  for nz in -1 1 2 5 9 13 14; do
    timeout 300 $R $BASE --gap $gap --noise $nz 2>&1 | tail -4 | sed "s/^RESULT/RESULT gap=$gap/" >> $M || echo "   (exit $?)" >> $M
  done
done
# beside the package's own conv kernels running in another process
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
for prec in 1 0; do
  python /tmp/conv_noise.py $prec 40 >> $M 2>&1 &
  NP=$!
  sleep 12
  for gap in 0 2; do
    timeout 300 $R --lanes 3 --F 4 --items 2400 --nf0 4 --nf1 2 --pattern war --gap $gap 2>&1 | tail -4 | sed "s/^RESULT/RESULT conv_process_prec=$prec gap=$gap/" >> $M
  done
  timeout 300 $R --lanes 3 --F 4 --items 2400 --nf0 4 --nf1 2 --pattern stemlds 2>&1 | tail -4 | sed "s/^RESULT/RESULT conv_process_prec=$prec/" >> $M
  wait $NP
done
grep -E "RESULT|words from|conv noise" $M | cut -c1-250
