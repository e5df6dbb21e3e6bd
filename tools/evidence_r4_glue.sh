# round 4: the rocprofv3 counter passes on the bandwidth-bound glue kernels again (warp / upsample / max-pool: unchanged since round 2, so this
# re-states profiles/r2_glue_pmc.txt on this round's library) -- north_star: "rocprof-reported HBM GB/s for the warp/upsample kernels"
set -x
R=$PWD; O=$R/gpurun_out/r4/glue; mkdir -p $O
python tools/glue_bench.py > $O/glue_hbm.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/A -o g --output-format csv -- python3 $R/tools/glue_bench.py > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/B -o g --output-format csv -- python3 $R/tools/glue_bench.py > $O/b.log 2>&1
cd $R
python tools/glue_pmc_summary.py $(find $O/A -name 'g_counter_collection.csv') $(find $O/B -name 'g_counter_collection.csv') > $O/glue_pmc.txt 2>&1
rm -rf $O/A $O/B
cat $O/glue_pmc.txt | head -40
