#!/usr/bin/env python3
"""Chooses, per 3x3 / stride-1 layer, between the direct MFMA conv and the Winograd forms F(2x2,3x3) / F(4x4,3x3) from per-op timing
tables of the SAME program recorded three times (bench.py --per-op under VIDC_WINOGRAD=0, 2, 4; HIP events between ops, rescaled to
the graph replay time), and writes the choice into conv_tuning.json as "W:<direct signature>": [m_fp32, m_mixed] (m = 0, 2 or 4).
A Winograd layer's time = wino_in + the grouped GEMM launch + wino_out.  CPU only.

    python tools/winograd_select.py --fp32 w0.tsv.fp32 w2.tsv.fp32 w4.tsv.fp32 --mixed w0.tsv w2.tsv w4.tsv [--min-gain 0.03]
"""
import argparse
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "vi_depth_completion_amd", "conv_tuning.json")


def layers(path):
    """{layer key: (us, direct signature or None)} of the conv layers of one per-op table; Winograd triples are summed."""
    rows = [ln.rstrip("\n").split("\t") for ln in open(path)]
    out = {}
    for i, (_p, us, name) in enumerate(rows):
        if not name.startswith("conv:"):
            continue
        key = name.split(":")[1]
        sig = name.split(" ")[1]
        if "@wino" in key:
            assert rows[i - 1][2].startswith("wino_in") and rows[i + 1][2].startswith("wino_out"), name
            out[key.split("@")[0]] = (float(rows[i - 1][1]) + float(us) + float(rows[i + 1][1]), None)
        else:
            out[key] = (float(us), sig)
    return out


def choose(files, min_gain):
    d, w2, w4 = (layers(f) for f in files)
    pick, total = {}, [0.0, 0.0]
    for key, (us0, sig) in d.items():
        if not re.search(r"_k3s1_", sig or ""):
            total[0] += us0; total[1] += us0
            continue
        cands = [(us0, 0)]
        for m, t in ((2, w2), (4, w4)):
            if key in t and t[key][1] is None:
                cands.append((t[key][0], m))
        best = min(cands)
        if best[1] and best[0] > (1.0 - min_gain) * us0:
            best = (us0, 0)
        prev = pick.get(sig)
        if prev is None or best[0] / us0 < prev[2]:          # several layers share a signature: all of them must agree -> keep the stronger verdict
            pick[sig] = (best[1], {m: "%.1f" % u for u, m in cands}, best[0] / us0)
        total[0] += us0; total[1] += best[0]
    return pick, total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fp32", nargs=3, required=True)
    ap.add_argument("--mixed", nargs=3)
    ap.add_argument("--min-gain", type=float, default=0.03)
    ap.add_argument("--dry", action="store_true")
    a = ap.parse_args()
    table = json.load(open(OUT))
    p32, t32 = choose(a.fp32, a.min_gain)
    pmx, tmx = choose(a.mixed, a.min_gain) if a.mixed else ({}, None)
    for sig in sorted(set(p32) | set(pmx)):
        m32 = p32[sig][0] if sig in p32 else 0
        mmx = pmx[sig][0] if sig in pmx else m32
        table["W:" + sig] = [m32, mmx]
        print("%-36s fp32 -> F%d %s   mixed -> F%d %s" % (sig, m32, p32.get(sig, ("", ""))[1], mmx, pmx.get(sig, ("", ""))[1]))
    print("conv time per tick, fp32: %.0f -> %.0f us" % tuple(t32) + ("; mixed: %.0f -> %.0f us" % tuple(tmx) if tmx else ""))
    if not a.dry:
        json.dump(dict(sorted(table.items())), open(OUT, "w"), indent=0)
        print("updated", OUT)


if __name__ == "__main__":
    main()
