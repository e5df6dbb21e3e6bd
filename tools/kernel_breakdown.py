#!/usr/bin/env python3
"""Per-tick kernel table from rocprofv3 `--kernel-trace [--stats]` output.

    python tools/kernel_breakdown.py KERNEL_TRACE.csv TICKS > out.txt        per-dispatch trace: calls/tick, MEDIAN, mean, max, us/tick
    python tools/kernel_breakdown.py KERNEL_STATS.csv TICKS > out.txt        (--stats summary only: mean instead of median)
    python tools/kernel_breakdown.py KERNEL_TRACE.csv 100 warp_fwd_kernel 25  steady-state window: only the dispatches between the
                                                                              25th and the 125th launch of `warp_fwd_kernel` (one per tick),
                                                                              so one-off work (weight upload / packing: ~5600 copies) is out

TICKS = executions of the timed program in the traced process (for bench.py: steps + warmup + 1 drain tick per run() call + the
capture warm-up + the eager / graph timing passes of the roofline section; printed by the caller).  The median is what to read: one
25 ms outlier in 5000 launches moved a mean by 30 % in round 1."""
import csv
import statistics
import sys


def clean(name):
    return name.replace("(anonymous namespace)::", "")


def skip(name):
    return name.startswith("void at::") or "elementwise_kernel" in name      # torch's one-off parameter initialisation


def main():
    path, ticks = sys.argv[1], float(sys.argv[2])
    rows = list(csv.DictReader(open(path)))
    total = 0.0
    if rows and "Start_Timestamp" in rows[0]:
        if len(sys.argv) > 4:
            marker, skip_n = sys.argv[3], int(sys.argv[4])
            marks = sorted(int(r["Start_Timestamp"]) for r in rows if marker in r["Kernel_Name"])
            lo, hi = marks[skip_n], marks[skip_n + int(ticks)]
            rows = [r for r in rows if lo <= int(r["Start_Timestamp"]) < hi]
            print("# window: launches %d..%d of %s = %.0f ticks, %.3f ms per tick wall" % (skip_n, skip_n + int(ticks), marker, ticks, (hi - lo) / 1e6 / ticks))
        per = {}
        for r in rows:
            per.setdefault(clean(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        print("%-78s %10s %10s %10s %10s %10s" % ("kernel", "calls/tick", "median_us", "mean_us", "max_us", "us/tick"))
        for name, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            if skip(name):
                continue
            print("%-78s %10.1f %10.2f %10.2f %10.1f %10.1f" % (name[:78], len(d) / ticks, statistics.median(d), sum(d) / len(d), max(d), sum(d) / ticks))
            total += sum(d) / ticks
        t0 = min(int(r["Start_Timestamp"]) for r in rows)
        t1 = max(int(r["End_Timestamp"]) for r in rows)
        print("total: %.1f us of kernel time per tick (%d dispatches over %.1f ms of trace)" % (total, len(rows), (t1 - t0) / 1e6))
        return
    print("%-78s %10s %10s %10s" % ("kernel", "calls/tick", "avg_us", "us/tick"))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        name = clean(r["Name"])
        if skip(name):
            continue
        calls, tot = int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3
        print("%-78s %10.1f %10.2f %10.1f" % (name[:78], calls / ticks, float(r["AverageNs"]) / 1e3, tot / ticks))
        total += tot / ticks
    print("total: %.1f us per tick" % total)


if __name__ == "__main__":
    main()
