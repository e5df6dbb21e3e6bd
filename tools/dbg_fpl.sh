#!/bin/bash
# the lane-independence check under a few switches (debugging aid)
cd "$(dirname "$0")/.."
for cfg in "" "VIDC_WINOGRAD=0" "VIDC_EXEC=eager" "VIDC_NO_BUFFER_REUSE=1" "VIDC_WINOGRAD=0 VIDC_EXEC=eager"; do
  echo "== F=2 fp32 [$cfg]"
  env $cfg python tools/dbg_fpl.py 2 fp32 2>&1 | grep -v amdgpu.ids | tail -12
done
echo "== F=4 fp32"; python tools/dbg_fpl.py 4 fp32 2>&1 | grep -v amdgpu.ids | tail -12
echo "== F=2 mixed"; python tools/dbg_fpl.py 2 mixed 2>&1 | grep -v amdgpu.ids | tail -12
