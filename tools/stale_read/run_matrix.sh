#!/bin/bash
# Runs tools/stale_read/repro over the switches VERDICT r5 item 1(a) names; one RESULT line per configuration in gpurun_out/stale/matrix.txt,
# plus the runtime's own AQL packet log (AMD_LOG_LEVEL=4) of a short run, reduced to the packet headers (acquire / release scopes).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/stale
mkdir -p $OUT
R=tools/stale_read/repro
M=$OUT/matrix.txt
: > $M
run() {   # run <label> <env...> -- <args...>
    local label="$1"; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    echo "## $label   env: ${envs[*]:-}   args: $*" >> $M
    env "${envs[@]}" timeout 300 $R "$@" >> $M 2>&1 || echo "   (exit $?)" >> $M
}
BASE="--lanes 3 --F 4 --items 2000"
for rep in 1 2 3; do run "base rep $rep" -- $BASE; done
run "eager" -- $BASE --exec eager
run "hwq1" GPU_MAX_HW_QUEUES=1 -- $BASE
run "hwq2" GPU_MAX_HW_QUEUES=2 -- $BASE
run "hwq8" GPU_MAX_HW_QUEUES=8 -- $BASE
run "no packet capture" DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 -- $BASE
run "img sys + rec sys" -- $BASE --img sys --rec sys
run "img inv (buffer_inv sc0 sc1 at entry)" -- $BASE --img inv
run "img nontemporal" -- $BASE --img nt
run "rec vector" -- $BASE --rec vector
run "rec scalar + s_dcache_inv" -- $BASE --rec dcinv
run "band pattern (warp_fwd shape)" -- $BASE --pattern band
run "band pattern rep 2" -- $BASE --pattern band
run "copies by kernel" -- $BASE --copy kernel
run "host sync before launch" -- $BASE --hostsync 1
run "lanes 1" -- --lanes 1 --F 4 --items 2000
run "lanes 2" -- --lanes 2 --F 4 --items 2000
run "lanes 4" -- --lanes 4 --F 4 --items 2000
run "F 1" -- --lanes 3 --F 1 --items 2000
run "no fillers" -- $BASE --nf0 0 --nf1 0
run "few fillers" -- $BASE --nf0 8 --nf1 4
run "many fillers" -- $BASE --nf0 200 --nf1 100
run "small fillers" -- $BASE --fill-mb 2 --nf0 150 --nf1 60
run "serialize kernel" AMD_SERIALIZE_KERNEL=3 -- $BASE
run "serialize copy" AMD_SERIALIZE_COPY=3 -- $BASE
run "force blit copies" GPU_FORCE_BLIT_COPY_SIZE=65536 -- $BASE
# the packet headers the runtime writes
AMD_LOG_LEVEL=4 AMD_LOG_LEVEL_FILE=$OUT/clr_log timeout 300 $R --lanes 3 --F 4 --items 36 --nf0 4 --nf1 2 > $OUT/clr_run.txt 2>&1
for f in $OUT/clr_log*; do
    [ -f "$f" ] || continue
    grep -E "Dispatch Header|Barrier(AND|Value) Header|hipMemcpyAsync|hipGraphLaunch|Copy|SDMA|Blit|ShaderName" "$f" | sed -E 's/kernarg_address=0x[0-9a-f]+, //; s/correlation_id=[0-9]+, //' | tail -n 1500 > $OUT/clr_headers_$(basename $f).txt
    rm -f "$f"
done
AMD_LOG_LEVEL=4 AMD_LOG_LEVEL_FILE=$OUT/clr_log_eager timeout 300 $R --lanes 3 --F 4 --items 36 --nf0 4 --nf1 2 --exec eager > $OUT/clr_run_eager.txt 2>&1
for f in $OUT/clr_log_eager*; do
    [ -f "$f" ] || continue
    grep -E "Dispatch Header|Barrier(AND|Value) Header|ShaderName" "$f" | sed -E 's/kernarg_address=0x[0-9a-f]+, //; s/correlation_id=[0-9]+, //' | tail -n 800 > $OUT/clr_headers_eager_$(basename $f).txt
    rm -f "$f"
done
cat $M | grep -E "^##|RESULT|stale|exit" | head -200
