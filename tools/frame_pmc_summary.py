#!/usr/bin/env python3
"""Per-kernel HBM-side traffic and matrix-pipe activity of the frame program IN THE FRAME (every layer's weights HBM-cold, real
neighbours), from rocprofv3 --pmc passes on tools/frame_replay.py (one pass per counter group, MI355X_MICROARCH.md):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -- python3 tools/frame_replay.py 20        -> A/f_counter_collection.csv
    rocprofv3 --kernel-trace --pmc WRITE_SIZE ...                                           -> B/...
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE   -> C/...
    python tools/frame_pmc_summary.py A/f_counter_collection.csv B/f_counter_collection.csv C/f_counter_collection.csv

FETCH_SIZE / WRITE_SIZE are in KB; FETCH is doubled (gfx950 tallies 128-byte requests of wide reads at 64 bytes).  The steady-state
launches of a kernel are those after the first tick (warm-up run + capture excluded by taking the last `ticks` repetitions)."""
import csv
import statistics
import sys
from collections import defaultdict


def load(path):
    per = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0]
        per[name][r["Counter_Name"]].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    return per


def main():
    f, w, c = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
    ticks = int(sys.argv[4]) if len(sys.argv) > 4 else 20      # replays of tools/frame_replay.py (its argument)
    rows = []
    for name in f:
        if name.startswith("at::") or name.startswith("__amd") or "elementwise" in name:
            continue                                   # model construction (torch), not the frame
        fe = f[name]["FETCH_SIZE"]
        per_tick = len(fe) / (ticks + 1.0)            # one eager warm-up run + `ticks` replays
        fe = fe[-int(round(per_tick * ticks)):] if per_tick >= 1 else fe
        wr = w.get(name, {}).get("WRITE_SIZE", [])
        wr = wr[-len(fe):]
        mf = c.get(name, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", [])[-len(fe):]
        sb = c.get(name, {}).get("SQ_BUSY_CYCLES", [])[-len(fe):]
        ga = c.get(name, {}).get("GRBM_GUI_ACTIVE", [])[-len(fe):]
        us = statistics.mean(t for _, t in fe)
        fk = 2.0 * statistics.mean(v for v, _ in fe)
        wk = statistics.mean(v for v, _ in wr) if wr else float("nan")
        # chip-wide matrix-pipe busy fraction: busy cycles summed over all SIMDs / (active clocks x 256 CUs x 4 SIMDs); GRBM_GUI_ACTIVE is
        # summed over the 8 XCDs, hence the / 8
        busy = (sum(v for v, _ in mf) / max(sum(v for v, _ in ga) / 8.0 * 1024.0, 1.0)) if mf and ga else float("nan")
        rows.append((us * len(fe) / ticks, name, len(fe) / ticks, us, fk, wk, (fk + (wk if wk == wk else 0.0)) * 1024 / (us * 1e-6) / 1e12, busy))
    rows.sort(reverse=True)
    print("%-44s %7s %9s %12s %11s %9s %12s" % ("kernel (per tick)", "calls", "avg us", "FETCHx2 KB", "WRITE KB", "TB/s", "MFMA busy"))
    for tot, name, calls, us, fk, wk, tbs, busy in rows[:24]:
        print("%-44s %7.1f %9.2f %12.1f %11.1f %9.2f %11.1f%%" % (name[:44], calls, us, fk, wk, tbs, 100.0 * busy))
    print("(durations under counter collection are longer than in a plain trace; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), summed over the launches)")


if __name__ == "__main__":
    main()
