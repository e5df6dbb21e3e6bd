#!/usr/bin/env python3
"""rocprofv3 counter summary of the bandwidth-bound kernels (north_star: "rocprof-reported HBM GB/s for the warp/upsample kernels").

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d A -o g --output-format csv -- python3 tools/glue_bench.py      (one pass per counter:
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d B -o g --output-format csv -- python3 tools/glue_bench.py       MI355X_MICROARCH.md §HBM)
    python tools/glue_pmc_summary.py A/.../g_counter_collection.csv B/.../g_counter_collection.csv > profiles/r2_glue_pmc.txt

Per kernel and launch geometry (= batch size / shape): median duration from the trace's own timestamps, FETCH_SIZE (KB, raw and x2: on
gfx950 the counter tallies the 128-byte requests of wide coalesced reads at 64 bytes), WRITE_SIZE (KB), and the HBM-side rate
(2 x FETCH + WRITE) / duration against the 8 TB/s peak.  Launches are those of tools/glue_bench.py (5 warm-up + 50 timed per case)."""
import csv
import statistics
import sys

KEEP = ("warp_fwd_kernel", "warp_inv_rot_norm_kernel", "upsample_kernel", "maxpool_kernel", "wino_in_kernel", "wino_out_kernel", "stem_conv_kernel")


def load(path, counter):
    per = {}
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if not name.startswith(KEEP) or r["Counter_Name"] != counter:
            continue
        key = (name, int(r["Grid_Size"]), int(r["Workgroup_Size"]))
        per.setdefault(key, []).append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    return per


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    print("%-26s %10s %8s %10s %12s %12s %12s %10s %8s" % ("kernel", "grid", "launches", "median_us", "FETCH_KB", "FETCHx2_KB", "WRITE_KB", "GB/s", "of 8TB/s"))
    for key in sorted(f):
        fv, wv = f[key], w.get(key, [])
        us = statistics.median([t for _, t in fv])
        fetch = statistics.median([v for v, _ in fv])
        write = statistics.median([v for v, _ in wv]) if wv else float("nan")
        gbs = (2 * fetch + write) * 1024 / us / 1e3
        print("%-26s %10d %8d %10.2f %12.1f %12.1f %12.1f %10.1f %7.1f%%" % (key[0], key[1], len(fv), us, fetch, 2 * fetch, write, gbs, 100 * gbs / 8000.0))


if __name__ == "__main__":
    main()
