#!/bin/bash
# second call: the round-5 scenario itself (mixed mode, F = 1, 7-frame streams over lanes 1,2,3,2,1) with the round-5 kernel form
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/stale
P=gpurun_out/stale/pipeline2.txt
: > $P
for rep in 1 2 3 4 5 6; do
  echo "## dbg_lanes rep $rep: VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=1 (mixed)" >> $P
  VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=1 timeout 300 python tools/dbg_lanes.py >> $P 2>&1; echo "   exit $?" >> $P
done
for rep in 1 2; do
  echo "## dbg_stem_race rep $rep: VIDC_DBG_STEM_LOADS=1" >> $P
  VIDC_DBG_STEM_LOADS=1 timeout 300 python tools/dbg_stem_race.py >> $P 2>&1; echo "   exit $?" >> $P
done
st() { echo "## env: $* args: $ARGS" >> $P; env "$@" timeout 900 python tools/stale_read/stress_pipeline.py $ARGS >> $P 2>&1; echo "   exit $?" >> $P; }
ARGS="--items 600 --runs 3 --F 1 --lanes 3"
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=1
ARGS="--items 600 --runs 3 --F 1 --lanes 2"
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=1
ARGS="--items 800 --runs 3 --F 4 --lanes 3"
st VIDC_PRECISION=mixed VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=1
grep -E "^##|STRESS|exit|differs|mismatch|rror" $P | tail -80
