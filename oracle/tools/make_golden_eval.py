"""Golden vectors for the evaluation figures, produced by the REFERENCE ITSELF: `network_run._network_evaluate`
(network_run.py:198-225) and the statistics block of `network_run.evaluate` (network_run.py:349-403) are imported from /root/reference
(build container only; oracle/tools/ref_shims.py) and run on small seeded inputs.  TEST INFRASTRUCTURE.

    python oracle/tools/make_golden_eval.py [out_dir]      ->  tests/golden/eval_reference.npz

Stored: the inputs (two batches: predicted / ground-truth normals, plane mask, predicted / ground-truth depth), the per-batch error
arrays `_network_evaluate` returned, and the figures the reference LOGGED ('NORMAL ERROR STATS: ...', 'DEPTH ERROR STATS: ...',
parsed from its own log lines, i.e. six decimals)."""
import logging
import os
import re
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import ref_shims  # noqa: E402


def make_batches(seed=11, n_batches=2, B=2, H=48, W=64):
    g = torch.Generator().manual_seed(seed)
    batches, outputs = [], []
    for _ in range(n_batches):
        gt_n = torch.nn.functional.normalize(torch.randn(B, 3, H, W, generator=g), dim=1) * (0.5 + torch.rand(B, 1, H, W, generator=g))   # not unit: the reference normalises
        pred_n = gt_n + 0.35 * torch.randn(B, 3, H, W, generator=g)
        mask = (torch.rand(B, H, W, generator=g) > 0.3).float() * (1 + (torch.rand(B, H, W, generator=g) > 0.5).float())               # 0, 1, 2
        gt_d = torch.rand(B, 1, H, W, generator=g) * 5.0
        gt_d = gt_d * (torch.rand(B, 1, H, W, generator=g) > 0.25).float()                                                             # invalid = 0
        pred_d = (gt_d + 0.3 * torch.randn(B, 1, H, W, generator=g)).abs() + 0.05 * torch.rand(B, 1, H, W, generator=g)
        pred_d[0, 0, 0, :4] = 0.0            # zero predictions: ratio inf / nan like in the reference
        batches.append({"image": torch.zeros(B, 3, H, W), "normal": gt_n, "mask": mask, "depth": gt_d})
        outputs.append((pred_n, pred_d))
    return batches, outputs


def main(out_dir):
    ref_shims.install()
    import network_run  # noqa: the reference's network_run.py

    batches, outputs = make_batches()
    iface = network_run.ImageNetworkRunInterface
    stub = types.SimpleNamespace(
        _network_estimates_normal=lambda: True, _network_estimates_depth=lambda: True,
        _get_network_output_normal=lambda out: out[0], _get_network_output_depth=lambda out: out[1],
        args=types.SimpleNamespace(save=""), test_dataloader=batches)
    per_batch = [iface._network_evaluate(stub, b, o) for b, o in zip(batches, outputs)]
    stub._run_evaluation_iteration = lambda sample, i: per_batch[i]

    lines = []

    class Grab(logging.Handler):
        def emit(self, record):
            lines.append(record.getMessage())

    root = logging.getLogger()
    h = Grab()
    root.addHandler(h)
    old = root.level
    root.setLevel(logging.INFO)
    try:
        iface.evaluate(stub)
    finally:
        root.removeHandler(h)
        root.setLevel(old)
    nline = next(l for l in lines if l.startswith("NORMAL ERROR STATS"))
    dline = next(l for l in lines if l.startswith("DEPTH ERROR STATS"))
    num = r"([-+0-9.eE]+|nan|inf)"
    nm = re.match(r"NORMAL ERROR STATS: Mean %s, Median %s, Rmse %s, 5deg %s, 7.5deg %s, 11.25deg %s, 22.5deg %s, 30deg %s" % ((num,) * 8), nline)
    dm = re.match(r"DEPTH ERROR STATS: MAD: %s, RMSE: %s, 1.05 %s 1.10: %s 1.25: %s 1.25\^2: %s, 1.25\^3: %s" % ((num,) * 7), dline)
    out = {"normal_figures": np.array([float(v) for v in nm.groups()]), "depth_figures": np.array([float(v) for v in dm.groups()]),
           "normal_log_line": np.array(nline), "depth_log_line": np.array(dline)}
    for i, (b, o, e) in enumerate(zip(batches, outputs, per_batch)):
        out.update({"b%d.gt_normal" % i: b["normal"].numpy(), "b%d.mask" % i: b["mask"].numpy(), "b%d.gt_depth" % i: b["depth"].numpy(),
                    "b%d.pred_normal" % i: o[0].numpy(), "b%d.pred_depth" % i: o[1].numpy(),
                    "b%d.normal_error" % i: e[0], "b%d.depth_ratio_error" % i: e[1], "b%d.depth_abs_error" % i: e[2]})
    path = os.path.join(out_dir, "eval_reference.npz")
    np.savez_compressed(path, **out)
    print(nline)
    print(dline)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden"))
