#!/bin/bash
# Attribution of the fused Winograd launch (csrc/wfused.hip): the same launch without its MFMAs / U loads / transform / fold
# (make wfused_attrib; VIDC_WFUSED_DBG bits 1 / 2 / 4 / 8 pick the instantiation)
cd "$(dirname "$0")/.."
[ -f vi_depth_completion_amd/libvidc_timing_wfused.so ] || make -C vi_depth_completion_amd/csrc wfused_attrib > /dev/null 2>&1
for dbg in ${1:-0 1 2 4 8 3 6 7 15}; do echo "== VIDC_WFUSED_DBG=$dbg"; VIDC_LIB_NAME=libvidc_timing_wfused.so VIDC_WFUSED_DBG=$dbg timeout 120 python tools/wfused_bench.py --iters 60 2>&1 | grep -E "layer3 conv2|one group" | cut -c1-150; done
