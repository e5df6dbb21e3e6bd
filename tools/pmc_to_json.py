#!/usr/bin/env python3
"""Folds the in-frame counter tables (tools/frame_pmc_summary.py output) and the per-op table of bench.py into profiles/pmc_traffic.json:
for every conv kernel instantiation of a leg -- per-launch FETCH_SIZE x 2 + WRITE_SIZE (HBM-side bytes), the matrix-pipe busy fraction, and
the ALGORITHMIC bytes of its average launch -- what the fused op has to move at 4 bytes per element: input pixels x Cin, weights, the
residual it adds (VIDC_RESIDUAL), the tensor it accumulates into (VIDC_ACCUM), the fp32 output unless VIDC_NO_F32_OUT and the split-bf16
image of the output if VIDC_SPLIT_OUT -- from the signatures and flags of the ops that run on it (the frame program recorded in dry-run
mode on the CPU, 320x256, batch VIDC_REPLAY_BATCH (default 4: four stream items per launch): the same op list the counters were taken on).  bench.py reports the entry of a leg's dominant kernel
as roofline.traffic.

    python tools/pmc_to_json.py profiles/r3_frame_pmc_mixed.txt mixed "rocprofv3 ... frame_replay.py 20" [--out profiles/pmc_traffic.json]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def frame_op_names(mode):
    import numpy as np
    import torch
    os.environ["VIDC_PRECISION"] = mode
    from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
    from vi_depth_completion_amd.networks.surface_normal import SurfaceNormalPrediction
    from vi_depth_completion_amd.pipeline import build_frame_program
    H = 256
    sn = SurfaceNormalPrediction(fc_img=np.array([202.0, 202.0]), cc_img=np.array([0.5 * 319.87654, 0.5 * 239.87603 * H / 240.0]), output_size=(H, 320)).eval()
    dc = ModifiedFPN().eval()
    return build_frame_program(sn, dc, int(os.environ.get("VIDC_REPLAY_BATCH", "4")), H, 320, torch.device("cpu"), dry_run=True).op_names


def main():
    pmc_txt, mode, command = sys.argv[1], sys.argv[2], sys.argv[3]
    prec = 0 if mode == "fp32" else 1
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(ROOT, "profiles", "pmc_traffic.json")
    import bench
    alg, cnt, shapes = {}, {}, {}
    for name in frame_op_names(mode):
        if not name.startswith("conv:"):
            continue
        _c, _key, tile, _sk, rest = name.split(":", 4)
        if (1 if rest.startswith("bf16x3 ") else 0) != prec:
            continue
        m = re.search(r"M(\d+)_N(\d+)_K(\d+)_k(\d)s(\d)_G(\d+)", name)
        M, N, K, k, s, G = (int(v) for v in m.groups())
        fl = int(re.search(r"flags=0x([0-9a-f]+)", name).group(1), 16)
        outs = (0 if fl & 128 else 1) + (1 if fl & 64 else 0) + (1 if fl & 8 else 0) + (1 if fl & 32 else 0)      # f32 out, split image, residual in, accumulate in
        kern = bench.kernel_name(tile, prec)
        alg[kern] = alg.get(kern, 0) + 4 * G * (M * s * s * (K // (k * k)) + N * K + outs * M * N)
        cnt[kern] = cnt.get(kern, 0) + 1
        shapes.setdefault(kern, {})
        shapes[kern][m.group(0)] = shapes[kern].get(m.group(0), 0) + 1
    table = json.load(open(out)) if os.path.exists(out) else {}
    table["_comment"] = ("HBM-side traffic per launch of conv kernel instantiations IN THE FRAME (tools/frame_replay.py: the frame program replayed 20 "
                         "times, every layer's weights HBM-cold), from separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES "
                         "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE) with --kernel-trace only; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies wide "
                         "coalesced reads at half).  algorithmic_bytes = activations in + weights + output of the kernel's average launch.  bench.py "
                         "reports the entry of a leg's dominant kernel as roofline.traffic.")
    for ln in open(pmc_txt):
        m = re.match(r"(conv_igemm_f32<[^>]*>)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.naN]+)\s+([\d.]+)\s+([\d.naN]+)%", ln)
        if not m:
            continue
        kern, calls, us, fk, wk, _tbs, busy = m.group(1), float(m.group(2)), float(m.group(3)), float(m.group(4)), m.group(5), m.group(6), m.group(7)
        if kern not in alg:
            continue
        wkb = float(wk) if wk.lower() != "nan" else 0.0
        table[kern] = {"round": int(os.environ.get("VIDC_ROUND", "5")), "program_batch": int(os.environ.get("VIDC_REPLAY_BATCH", "4")), "launches_per_tick": calls, "avg_us_under_counter_collection": us,
                       "fetch_bytes": int(fk * 1024), "write_bytes": int(wkb * 1024), "traffic_bytes": int((fk + wkb) * 1024),
                       "algorithmic_bytes": int(alg[kern] / cnt[kern]), "traffic_over_algorithmic": round((fk + wkb) * 1024 / (alg[kern] / cnt[kern]), 3),
                       "mfma_busy_fraction": (round(float(busy) / 100.0, 4) if busy.lower() != "nan" else None),
                       "shape": "the %d launches per tick of this instantiation: %s" % (cnt[kern], ", ".join("%d x %s" % (v, k) for k, v in sorted(shapes[kern].items(), key=lambda kv: -kv[1])[:4])),
                       "command": command}
    with open(out, "w") as f:
        json.dump(table, f, indent=1)
    print("wrote", out, "with", len(table) - 1, "kernels")


if __name__ == "__main__":
    main()
