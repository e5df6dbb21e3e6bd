"""Golden vector for the `use_mask=True` branch of the reference's SurfaceNormalPrediction (networks/surface_normal.py:150-162), produced by
the reference itself on the CPU (build container only).  TEST INFRASTRUCTURE.   -> tests/golden/sn_use_mask.npz
Stored: gravity / aligned direction of the frame (a strongly tilted one, so that the warp leaves empty borders and the mask matters),
the unit normals on a 4x4-subsampled grid + their sum."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shims  # noqa: E402
from vi_depth_completion_amd import synthetic as S  # noqa: E402

torch.set_grad_enabled(False)
ref = ref_shims.load_reference()
sn = ref.SurfaceNormalPrediction(fc_img=np.array([202.0, 202.0]), use_mask=True)
st = sn.state_dict()
st.update(S.seeded_state_dict(sn.state_dict(), 1234))
sn.load_state_dict(st)
sn.eval()
b = S.synthetic_batch(1, 240, 320, 1234, frame0=3)
g = torch.nn.functional.normalize(torch.tensor([[0.35, 0.9, 0.25]]), dim=1)
a = torch.tensor([[0.0, 1.0, 0.0]])
n = sn(b["image"], g, a)
sn.use_mask = False
n_off = sn(b["image"], g, a)
print("normals with mask vs without: max diff %.3e (the branch matters on this frame)" % float((n - n_off).abs().max()))
out = os.path.join(ROOT, "tests", "golden", "sn_use_mask.npz")
np.savez_compressed(out, gravity=g.numpy(), aligned=a.numpy(), frame0=np.int64(3), normals_sub=n[0, :, ::4, ::4].numpy(), normals_sum=np.float64(n.double().sum()),
                    diff_vs_unmasked=np.float64((n - n_off).abs().max()))
print("wrote", out, os.path.getsize(out))
