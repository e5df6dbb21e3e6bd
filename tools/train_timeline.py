#!/usr/bin/env python3
"""Where a captured training step spends its time, from a rocprofv3 `--kernel-trace` CSV of `tools/train_bench.py` (graph replay).

    python tools/train_timeline.py KERNEL_TRACE.csv [STEPS_FROM_END]

The last steps of the trace are found by their `adam_kernel` launches (one per step).  For the window between the last two of
them: wall time, union of busy intervals (all queues), per queue the number of kernels / busy time / idle gaps between consecutive
kernels, the histogram of those gaps, and the chip's concurrency profile (time with 0, 1, 2, 3+ kernels in flight)."""
import csv
import sys


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = [r for r in csv.DictReader(open(path))]
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].replace("(anonymous namespace)::", "")) for r in rows))
    adam = [s for s, e, q, n in ev if "adam_kernel" in n]
    lo, hi = adam[-1 - back], adam[-back]
    win = [x for x in ev if lo <= x[0] < hi]
    print("window: one step between two Adam launches = %.3f ms wall, %d kernels" % ((hi - lo) / 1e6, len(win)))
    # concurrency profile
    pts = sorted([(s, 1) for s, e, q, n in win] + [(min(e, hi), -1) for s, e, q, n in win])
    conc, last, level = {}, lo, 0
    for t, d in pts:
        conc[level] = conc.get(level, 0) + (t - last)
        last, level = t, level + d
    conc[level] = conc.get(level, 0) + (hi - last)
    tot = float(hi - lo)
    print("kernels in flight:  " + "  ".join("%d: %.2f ms (%.0f %%)" % (k, v / 1e6, 100 * v / tot) for k, v in sorted(conc.items()) if v > 0))
    print("sum of kernel durations %.2f ms" % (sum(min(e, hi) - s for s, e, q, n in win) / 1e6))
    per = {}
    for x in win:
        per.setdefault(x[2], []).append(x)
    for q, xs in sorted(per.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, _, _ in xs)
        gaps = [xs[i + 1][0] - xs[i][1] for i in range(len(xs) - 1)]
        pos = [g for g in gaps if g > 0]
        hist = [0] * 7
        for g in pos:
            us = g / 1e3
            hist[0 if us < 1 else 1 if us < 2 else 2 if us < 4 else 3 if us < 8 else 4 if us < 16 else 5 if us < 64 else 6] += 1
        print("queue %s: %5d kernels, busy %.2f ms, first..last %.2f ms, gaps>0: %d totalling %.2f ms (median %.2f us)  hist <1|<2|<4|<8|<16|<64|>=64 us: %s" % (
            q, len(xs), busy / 1e6, (xs[-1][1] - xs[0][0]) / 1e6, len(pos), sum(pos) / 1e6, (sorted(pos)[len(pos) // 2] / 1e3 if pos else 0), hist))
    # phases: forward = up to the loss kernel, decoder backward, pyramid backward
    marks = [(s, n) for s, e, q, n in win if "l1_loss" in n or "head_dgrad" in n or "head_wgrad_final" in n or "pack_batched" in n or "maxpool_bwd" in n]
    for s, n in marks:
        print("  mark %-28s at %.3f ms" % (n[:28], (s - lo) / 1e6))
    # the ten largest gaps with no kernel in flight anywhere
    idle, cur_end = [], lo
    for s, e, q, n in win:
        if s > cur_end:
            idle.append((s - cur_end, cur_end - lo, n))
        cur_end = max(cur_end, e)
    idle.sort(reverse=True)
    print("chip idle (no kernel anywhere): %.2f ms in %d gaps; largest: %s" % (sum(g for g, _, _ in idle) / 1e6, len(idle),
          "; ".join("%.1f us at %.2f ms before %s" % (g / 1e3, t / 1e6, n[:24]) for g, t, n in idle[:8])))


if __name__ == "__main__":
    main()
