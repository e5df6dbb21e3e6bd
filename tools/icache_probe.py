#!/usr/bin/env python3
"""Do the three conv shapes of a ResNet layer-3 bottleneck (35 + 25 + 24 launches per tick) run faster IN THE FRAME when they share one
kernel instantiation (instruction cache stays warm across consecutive launches) than with the per-shape winners of the isolated tuner
(three different instantiations cycling)?  Whole-tick hipGraph replay time of the frame program for a few joint assignments.

    python tools/icache_probe.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vi_depth_completion_amd import _lib as L, engine                       # noqa: E402
from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, build_frame_program      # noqa: E402

SIGS = ["M320_N256_K2304_k3s1_G4", "M320_N1024_K256_k1s1_G4", "M320_N256_K1024_k1s1_G4"]


def main():
    H, W, B = 256, 320, 1
    dev = torch.device("cuda")
    torch.set_grad_enabled(False)
    cc = (0.5 * 319.87654, 0.5 * 239.87603 * H / 240.0)
    pipe = DepthCompletionPipeline(enriched_samples=200, cc_img=cc, device=dev)
    table = engine.tuning_table()
    base = {s: list(table[s]) for s in SIGS}
    ws = engine.JointWeightStore({"sn": pipe.surface_normal_cnn, "dc": pipe.cnn})
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)

    def tick_ms():
        prog = build_frame_program(pipe.surface_normal_cnn, pipe.cnn, B, H, W, dev, weights=ws)
        prog.run()
        prog.capture_segments()
        prog.time(iters=3, use_graph=True)
        return min(prog.time(iters=20, use_graph=True) for _ in range(3))

    print("per-shape winners %s: %.3f ms" % ([(L.TILE_NAMES[base[s][0]], base[s][1]) for s in SIGS], tick_ms()), flush=True)
    for tile in (13, 18, 5, 4, 17, 20, 2, 23, 21, 6):
        for sks in ((3, 1, 1), (3, 1, 2), (3, 2, 2), (2, 1, 1)):
            for s, sk in zip(SIGS, sks):
                e = base[s]
                table[s] = [tile, sk, 1, e[3], e[4]] if len(e) >= 5 else [tile, sk, 1]
            try:
                ms = tick_ms()
            except RuntimeError as ex:
                print("tile %s sk %s: failed (%s)" % (L.TILE_NAMES[tile], sks, str(ex)[:60]))
                continue
            print("all three on %-10s split-K %s: %.3f ms" % (L.TILE_NAMES[tile], sks, ms), flush=True)
    for s in SIGS:
        table[s] = base[s]


if __name__ == "__main__":
    main()
