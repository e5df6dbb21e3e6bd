#!/usr/bin/env python3
"""Time of the two bandwidth-bound launches at the ends of a training step (ModifiedFPN, 310 M parameters): the re-packing of every conv
weight (`DepthCompletionTrainer.repack`, one launch) and the Adam update (one launch), HIP events over 20 repetitions each.
    VIDC_TRAIN_PRECISION=bf16 python tools/pack_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vi_depth_completion_amd import _lib as L, synthetic as S  # noqa: E402
from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN  # noqa: E402
from vi_depth_completion_amd.training import DepthCompletionTrainer  # noqa: E402


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dev = torch.device("cuda")
    torch.set_grad_enabled(False)
    cnn = ModifiedFPN().to(dev)
    cnn.load_state_dict(S.seeded_state_dict(cnn.state_dict(), 1234, device=dev))
    cnn.train()
    tr = DepthCompletionTrainer(cnn, 1e-4)
    b = S.synthetic_batch(2, 240, 320, 1234)
    image = b["image"].to(dev)
    normal = torch.nn.functional.normalize(image - 0.5, dim=1)
    gt = S.synthetic_ground_truth_depth(b["image"], 1234).to(dev)
    tr.forward_backward(image, normal, b["sparse_depth"].to(dev), gt)          # creates the packed copies and the table
    n = tr.flat_p.numel()
    elems = sum(it[0].numel() for it in tr._pack_items)
    e = 2 if tr.precision == L.PREC_BF16 else 4
    t = timed(tr.repack)
    print("repack: %d items, %.1f M elements: %.3f ms = %.2f TB/s (4 B read + %d B written per element)" % (len(tr._pack_items), elems / 1e6, t, elems * (4 + e) / t / 1e9, e))
    t = timed(lambda: L.check(L.lib().vidc_adam_step(L.ptr(tr.flat_p), L.ptr(tr.flat_g), L.ptr(tr.m), L.ptr(tr.v), n, 0.0, 0.9, 0.999, 1e-8, 1, L.current_stream()), "adam"))
    print("adam (lr 0): %.1f M parameters: %.3f ms = %.2f TB/s (16 B read + 12 B written per parameter)" % (n / 1e6, t, n * 28 / t / 1e9))


if __name__ == "__main__":
    main()
