"""Where the time of the persistent chain kernel goes: per-item stamps (vidc_chain_trace) of the frame program's chain, summarised per
layer: wall time of the layer on each XCD, dependency wait, main loop, epilogue+drain, done-count latency."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["VIDC_CHAIN"] = "1"          # the chain is opt-in
sys.path.insert(0, ROOT)
from vi_depth_completion_amd import _lib as L, synthetic as S  # noqa: E402
from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN  # noqa: E402
from vi_depth_completion_amd.networks.surface_normal import SurfaceNormalPrediction  # noqa: E402
from vi_depth_completion_amd.pipeline import build_frame_program  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda")
sn = SurfaceNormalPrediction(fc_img=np.array([202.0, 202.0])).to(dev).eval()
dc = ModifiedFPN().to(dev).eval()
sn.load_state_dict(S.seeded_state_dict(sn.state_dict(), 1234, device=dev))
dc.load_state_dict(S.seeded_state_dict(dc.state_dict(), 1234, device=dev))
prog = build_frame_program(sn, dc, 1, 240, 320, dev)
ci = next(i for i, n in enumerate(prog.op_names) if n.startswith("chain"))
h = prog._chains[0]
prog.run()
prog.check_chains()
words = L.lib().vidc_chain_trace(h, 1, None, 0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    prog.run()          # whole program: the weights of the chain are HBM-cold when it starts, like in the frame
torch.cuda.synchronize()
e0.record()
L.check(L.lib().vidc_program_run_range(prog.handle, L.current_stream(), ci, ci + 1), "run_range")
e1.record()
torch.cuda.synchronize()
print("chain op alone: %.1f us" % (1e3 * e0.elapsed_time(e1)))
prog.run()
torch.cuda.synchronize()
buf = np.zeros(words, dtype=np.int64)
L.check(min(0, L.lib().vidc_chain_trace(h, 1, buf.ctypes.data, words)), "trace")
t = buf.reshape(-1, 160, 8)
valid = t[:, :, 0] > 0
print("workgroups with items:", int(valid.any(axis=1).sum()), "items traced:", int(valid.sum()))
rows = t[valid]                              # (n, 8)
layer = (rows[:, 6] >> 32).astype(int)
t0 = rows[:, 0].min()
us = lambda x: (x - t0) / 100.0
print("kernel span (first item start -> last done): %.1f us" % us(rows[:, 5].max()))
print("layer  items  start_us  wall_us | per item (us): claim->dep  dep wait  loop  epi+drain  count")
nl = layer.max() + 1
for l in list(range(0, 12)) + list(range(nl - 11, nl)):
    r = rows[layer == l]
    if len(r) == 0:
        continue
    print("%4d  %5d  %8.1f  %7.1f | %8.2f %8.2f %8.2f %8.2f %8.2f" % (
        l, len(r), us(r[:, 0].min()), (r[:, 5].max() - r[:, 0].min()) / 100.0, np.mean(r[:, 1] - r[:, 0]) / 100.0, np.mean(r[:, 2] - r[:, 1]) / 100.0,
        np.mean(r[:, 3] - r[:, 2]) / 100.0, np.mean(r[:, 4] - r[:, 3]) / 100.0, np.mean(r[:, 5] - r[:, 4]) / 100.0))
tot = {k: float(np.sum(rows[:, b] - rows[:, a])) / 100.0 for k, (a, b) in {"claim->dep": (0, 1), "dep wait": (1, 2), "loop": (2, 3), "epi+drain": (3, 4), "count": (4, 5)}.items()}
print("sum over items (us of workgroup time):", {k: round(v, 1) for k, v in tot.items()})
