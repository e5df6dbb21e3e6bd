#!/usr/bin/env python3
"""Per-kernel table of ONE captured training step from a rocprofv3 `--kernel-trace` CSV of `bench.py --train` / `tools/train_bench.py`
(the window between the last two Adam launches): launches, total and average duration per kernel family.  Under the tracer the step's
kernels run one after the other (its wall time is the SUM of the durations, not the concurrent step time).
    python tools/train_step_kernels.py KERNEL_TRACE.csv"""
import collections
import csv
import re
import sys


def main():
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    adam = [int(r["Start_Timestamp"]) for r in rows if "adam_kernel" in r["Kernel_Name"]]
    lo, hi = adam[-2], adam[-1]
    agg = collections.defaultdict(lambda: [0, 0])
    n = 0
    for r in rows:
        s = int(r["Start_Timestamp"])
        if not lo <= s < hi:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = re.sub(r"\(.*", "", name)
        if "at::native" in r["Kernel_Name"]:
            name = "torch: " + name[:60]
        agg[name][0] += 1
        agg[name][1] += int(r["End_Timestamp"]) - s
        n += 1
    tot = sum(v[1] for v in agg.values())
    print("one step between two Adam launches: %d kernels, %.2f ms of kernel time, %.2f ms wall under the tracer" % (n, tot / 1e6, (hi - lo) / 1e6))
    print("%-72s %6s %10s %9s" % ("kernel", "calls", "total ms", "avg us"))
    for name, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-72s %6d %10.3f %9.1f" % (name[:72], c, t / 1e6, t / c / 1e3))


if __name__ == "__main__":
    main()
