#!/bin/bash
# bisect: the mixed-mode reproducer (15 % of the items differ with the round-5 kernel form) against every switch
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/stale
P=gpurun_out/stale/pipeline3.txt
: > $P
st() { echo "## env: $* args: $ARGS" >> $P; env "$@" timeout 900 python tools/stale_read/stress_pipeline.py $ARGS >> $P 2>&1; echo "   exit $?" >> $P; }
ARGS="--items 400 --runs 2 --F 4 --lanes 3"
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=0
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=0
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=3
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=1
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=2
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=7
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=3 GPU_MAX_HW_QUEUES=1
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=3 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=3 VIDC_EXEC=eager
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=3 VIDC_FUSE_SPLIT=0
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=3 VIDC_WINOGRAD=0
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=3 VIDC_NO_BUFFER_REUSE=1
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=3 VIDC_TICK_VARIANTS=0
ARGS="--items 400 --runs 2 --F 4 --lanes 1"
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=3
ARGS="--items 400 --runs 2 --F 1 --lanes 3"
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=0
st VIDC_PRECISION=fp32 VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=3
ARGS="--items 400 --runs 2 --F 2 --lanes 2"
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=0
grep -E "^##|STRESS|exit|rror" $P | tail -120
