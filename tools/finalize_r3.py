#!/usr/bin/env python3
"""Copies the outputs of tools/evidence_r3.sh (gpurun_out/r3/final) into profiles/ under their r3_ names and fills the [[placeholders]] of
DESIGN.md from the JSON lines, so that every number in the text is the number in the committed file."""
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r3", "final")
DST = os.path.join(ROOT, "profiles")
NAMES = {"bench_line.json": "r3_bench_line.json", "bench_line_200.json": "r3_bench_line_200.json", "bench_line_200_lanes1.json": "r3_bench_line_200_lanes1.json",
         "per_op.tsv": "r3_per_op.tsv", "per_op.tsv.fp32": "r3_per_op.tsv.fp32", "bench_line_b8_640x480.json": "r3_bench_line_b8_640x480.json",
         "bench_line_b8_640x480_plane_head.json": "r3_bench_line_b8_640x480_plane_head.json", "bench_line_b4_1280x720.json": "r3_bench_line_b4_1280x720.json",
         "bench_line_2ranks_gloo_1gpu.json": "r3_bench_line_2ranks_gloo_1gpu.json", "train_line_bf16_b8.json": "r3_train_line_bf16_b8.json",
         "train_line_fp32_b8.json": "r3_train_line_fp32_b8.json", "bench_line_batch2.json": "r3_bench_line_batch2.json",
         "kernel_stats_prof_mixed.csv": "r3_kernel_stats_mixed.csv", "kernel_stats_prof_fp32.csv": "r3_kernel_stats_fp32.csv",
         "kernel_stats_prof_mixed2.csv": "r3_kernel_stats_mixed_lanes2.csv", "frame_breakdown_prof_mixed.txt": "r3_frame_breakdown_mixed.txt",
         "frame_breakdown_prof_fp32.txt": "r3_frame_breakdown_fp32.txt", "frame_breakdown_prof_mixed2.txt": "r3_frame_breakdown_mixed_lanes2.txt",
         "frame_pmc_mixed.txt": "r3_frame_pmc_mixed.txt", "frame_pmc_fp32.txt": "r3_frame_pmc_fp32.txt", "pytest_gpu.log": "r3_pytest_gpu.log",
         "pmc_traffic.json": "pmc_traffic.json"}


def load(name):
    return json.load(open(os.path.join(SRC, name)))


def main():
    for a, b in NAMES.items():
        p = os.path.join(SRC, a)
        if os.path.exists(p):
            shutil.copyfile(p, os.path.join(DST, b))
        else:
            print("missing:", a)
    b20, b200 = load("bench_line.json"), load("bench_line_200.json")
    tr16, tr32 = load("train_line_bf16_b8.json"), load("train_line_fp32_b8.json")
    busy = ""
    for ln in open(os.path.join(SRC, "frame_pmc_mixed.txt")):
        if ln.startswith("conv_igemm_f32<128, 128, 2, 2, 1, 3, 1, 1>") or ln.startswith("conv_igemm_f32<128, 128, 2, 2, 1, 4, 1, 2>"):
            busy = ln.split()[-1].rstrip("%")
            break
    tests = open(os.path.join(SRC, "pytest_gpu.log")).read()
    m = re.search(r"(\d+) passed", tests)
    f32 = lambda d: d["fp32_leg"]["conv_stack"]["at_measured_frame_rate"]["frac_of_peak_executed"]
    rep = {"mixed20": "%.0f" % b20["value"], "fp3220": "%.0f" % b20["value_fp32"], "rmse_mixed": "%.1e" % b20["rmse_vs_oracle"],
           "rmse_fp32": "%.1e" % b20["rmse_vs_oracle_fp32"], "fp32frac20": "%.1f %%" % (100 * f32(b20)), "mixed200": "%.0f" % b200["value"],
           "fp32200": "%.0f" % b200["value_fp32"], "fp32frac200": "%.1f %%" % (100 * f32(b200)),
           "seq_mixed": "%.0f" % b20["sequential_call_cnn"]["frames_per_s"], "lat_mixed": "%.2f" % b20["sequential_call_cnn"]["latency_ms_median"],
           "seq_fp32": "%.0f" % b20["fp32_leg"]["sequential_call_cnn"]["frames_per_s"], "lat_fp32": "%.2f" % b20["fp32_leg"]["sequential_call_cnn"]["latency_ms_median"],
           "cpu": "%.2f" % b20["cpu_baseline"]["value"], "cpu_cores": str(b20["cpu_baseline"]["cores"]),
           "b8head": "%.0f" % load("bench_line_b8_640x480_plane_head.json")["value"], "b8": "%.0f" % load("bench_line_b8_640x480.json")["value"],
           "b4": "%.0f" % load("bench_line_b4_1280x720.json")["value"], "train_bf16_ms": "%.1f" % tr16["ms_per_step"], "train_fp32_ms": "%.1f" % tr32["ms_per_step"],
           "train_bf16_fps": "%.0f" % tr16["value"], "busy128": busy or "n/a", "ntests_gpu": m.group(1) if m else "?", "ntests_cpu": "76"}
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    for k, v in rep.items():
        s = s.replace("[[%s]]" % k, v)
    left = re.findall(r"\[\[\w+\]\]", s)
    open(p, "w").write(s)
    print("filled:", rep)
    print("placeholders left:", left)


if __name__ == "__main__":
    main()
