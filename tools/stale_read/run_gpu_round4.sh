#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/stale
P=gpurun_out/stale/diag4.txt
: > $P
for loads in 3 2; do
  echo "## diag VIDC_DBG_STEM_LOADS=$loads mixed" >> $P
  VIDC_NO_BUFFER_REUSE=1 VIDC_PRECISION=mixed VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=$loads timeout 600 python tools/stale_read/diag_stem.py --items 360 >> $P 2>&1; echo "   exit $?" >> $P
done
echo "## diag VIDC_DBG_STEM_LOADS=3 fp32" >> $P
VIDC_NO_BUFFER_REUSE=1 VIDC_PRECISION=fp32 VIDC_FUSE_WARP=1 VIDC_DBG_STEM_LOADS=3 timeout 600 python tools/stale_read/diag_stem.py --items 360 >> $P 2>&1; echo "   exit $?" >> $P
tail -80 $P | cut -c1-400
