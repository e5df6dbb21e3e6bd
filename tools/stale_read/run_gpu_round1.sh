#!/bin/bash
# first GPU call of round 6: the stand-alone matrix, then the package-level stress with the round-5 failing form as control
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/stale
bash tools/stale_read/run_matrix.sh > gpurun_out/stale/matrix_stdout.txt 2>&1
P=gpurun_out/stale/pipeline.txt
: > $P
st() { echo "## env: $*" >> $P; env "$@" timeout 600 python tools/stale_read/stress_pipeline.py --items 800 --runs 2 >> $P 2>&1; echo "   exit $?" >> $P; }
st VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=1
st VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=1 GPU_MAX_HW_QUEUES=1
st VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
st VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=1 VIDC_EXEC=eager
st VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=2
st VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=0
st VIDC_FUSE_WARP=0
st VIDC_FUSE_WARP=0 GPU_MAX_HW_QUEUES=8
grep -E "^##|STRESS|exit|Error|error" $P | tail -60
