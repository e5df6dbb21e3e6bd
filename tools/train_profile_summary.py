#!/usr/bin/env python3
"""Per-kernel totals of ONE steady-state training step out of a rocprofv3 --kernel-trace database (rocpd sqlite) of tools/train_bench.py:
the window between the last two adam_kernel launches.

    python tools/train_profile_summary.py gpurun_out/r2/trainprof2/tp_results.db [rows]
"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    ad = db.execute("select start, end from kernels where name like '%adam_kernel%' order by start").fetchall()
    a0, a1 = ad[-2][1], ad[-1][1]
    rows = db.execute("select name, start, end from kernels where start >= ? and end <= ? order by start", (a0, a1)).fetchall()
    busy = sum(e - s for _, s, e in rows)
    print("step window %.2f ms, %d kernel launches, GPU busy %.2f ms" % ((a1 - a0) / 1e6, len(rows), busy / 1e6))
    agg = defaultdict(lambda: [0, 0])
    for n, s, e in rows:
        n = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")
        n = n.split("(")[0][:90]
        agg[n][0] += 1
        agg[n][1] += e - s
    print("%6s %10s %6s %9s  kernel" % ("calls", "total ms", "%", "avg us"))
    for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
        print("%6d %10.3f %6.1f %9.1f  %s" % (c, t / 1e6, 100.0 * t / busy, t / c / 1e3, n))


if __name__ == "__main__":
    main()
