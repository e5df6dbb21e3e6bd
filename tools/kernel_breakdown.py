#!/usr/bin/env python3
"""Per-tick kernel table from a rocprofv3 `--kernel-trace --stats` CSV:  python tools/kernel_breakdown.py STATS.csv TICKS > out.txt
TICKS = executions of the timed program in the traced process (for bench.py: steps + warmup + 1 drain tick per run() call + the
capture warm-up + the eager / graph timing passes of the roofline section; printed by the caller)."""
import csv
import sys


def main():
    path, ticks = sys.argv[1], float(sys.argv[2])
    rows = list(csv.DictReader(open(path)))
    print("%-78s %10s %10s %10s" % ("kernel", "calls/tick", "avg_us", "us/tick"))
    total = 0.0
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        name = r["Name"].replace("(anonymous namespace)::", "")
        if name.startswith("void at::") or "elementwise_kernel" in name:
            continue                                     # torch's one-off parameter initialisation
        calls, tot = int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3
        print("%-78s %10.1f %10.2f %10.1f" % (name[:78], calls / ticks, float(r["AverageNs"]) / 1e3, tot / ticks))
        total += tot / ticks
    print("total: %.1f us per tick" % total)


if __name__ == "__main__":
    main()
