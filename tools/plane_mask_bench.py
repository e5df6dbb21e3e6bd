#!/usr/bin/env python3
"""Times the plane-mask detector on the GPU: whole `run_on_batch`, its three engine programs (hipGraph replay) and
the kernels between them.      python tools/plane_mask_bench.py [--batches 1,8] [--height 240] [--per-op FILE]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vi_depth_completion_amd import synthetic as S                      # noqa: E402
from vi_depth_completion_amd.plane_mask import PlaneMaskDetector         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,8")
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--per-op", default="")
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    det = PlaneMaskDetector(device="cuda")
    det.load_state_dict(S.seeded_detector_state_dict(det.state_dict(), 1234, device="cuda"))
    for B in [int(v) for v in a.batches.split(",")]:
        img = torch.stack([S.uniform01(1234, "pmb%d" % i, (3, a.height, 320)) for i in range(B)]).cuda()
        for _ in range(3):
            ids = det.run_on_batch(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            ids = det.run_on_batch(img)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / a.iters
        dense, box, mask = det.model.programs(B, a.height, 320, det.device)
        for p in (dense, box, mask):          # run_on_batch replays ONE graph of the whole detector; the per-program graphs are for this table
            type(det.model)._execute(p)
        torch.cuda.synchronize()
        parts = {n: p.time(iters=20, use_graph=True) for n, p in (("dense", dense), ("box", box), ("mask", mask))}
        gf = {n: p.flops / 1e9 for n, p in (("dense", dense), ("box", box), ("mask", mask))}
        print("batch %d: run_on_batch %.3f ms = %.1f images/s; programs (graph replay) dense %.3f ms (%.1f GFLOP) box %.3f ms (%.1f) mask %.3f ms "
              "(%.1f); kernels in between + host %.3f ms; planes found %s" % (
                  B, ms, B * 1e3 / ms, parts["dense"], gf["dense"], parts["box"], gf["box"], parts["mask"], gf["mask"],
                  ms - sum(parts.values()), [int(v) for v in ids.flatten(1).max(1).values.cpu()]), flush=True)
        if a.per_op:
            with open(a.per_op + ".b%d" % B, "w") as f:
                for name, p in (("dense", dense), ("box", box), ("mask", mask)):
                    total, per = p.time(iters=5, use_graph=False, per_op=True)
                    for n, t in zip(p.op_names, per):
                        f.write("%s\t%.2f\t%s\n" % (name, t * 1e3, n))


if __name__ == "__main__":
    main()
