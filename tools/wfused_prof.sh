cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for dbg in 0 15; do
VIDC_WFUSED_DBG=$dbg rocprofv3 --kernel-trace --stats -d gpurun_out/wf_prof_$dbg -o wf -- python3 tools/wfused_bench.py --iters 30 > gpurun_out/wf_prof_$dbg.log 2>&1
f=$(find gpurun_out/wf_prof_$dbg -name "*kernel_stats.csv" | head -1)
echo "== dbg $dbg"; head -8 $f | cut -c1-200
done
