"""Debug/report: per-tensor error of the HIP training step against the reference's fixture (tests/golden/train_step.npz)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vi_depth_completion_amd import synthetic as S
from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
from vi_depth_completion_amd.training import DepthCompletionTrainer
torch.set_grad_enabled(False)
DEV = "cuda"
G = os.path.join(ROOT, "tests", "golden")
f = np.load(os.path.join(G, "train_step.npz"))
man = np.load(os.path.join(G, "state_dict_manifest.npz"))
shapes = {k: torch.empty(eval(s), device="meta") for k, s in zip(man["dc_keys"], man["dc_shapes"])}
sd = S.seeded_state_dict(shapes, 1234)
batch = S.synthetic_batch(2, 240, 320, 1234, frame0=int(f["frame0"]))
gt = S.synthetic_ground_truth_depth(batch["image"], 1234)
din = torch.zeros(2, 240, 320); rc = torch.from_numpy(f["depth_in_rc"]).long(); din[rc[:, 0], rc[:, 1], rc[:, 2]] = torch.from_numpy(f["depth_in_val"])
cnn = ModifiedFPN().to(DEV)
st = cnn.state_dict(); st.update({k: v.to(DEV) for k, v in sd.items()}); cnn.load_state_dict(st); cnn.train()
tr = DepthCompletionTrainer(cnn, float(f["lr"]))
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
loss, pred = tr.forward_backward(batch["image"].to(DEV), torch.from_numpy(f["normal"]).to(DEV), din[:, None].to(DEV), gt.to(DEV))
torch.cuda.synchronize(); t1 = time.perf_counter()
print("loss %.8f (ref %.8f)  pred probe max diff %.2e  first step wall %.1f ms" % (float(loss), float(f["loss"]), np.abs(pred[:, 0, ::16, ::16].cpu().numpy() - f["pred_probe"]).max(), 1e3 * (t1 - t0)))
for k in sorted({k.split("|")[1] for k in f.files if k.startswith("grad|")}):
    key = "grad|%s|" % k
    t = tr.grad[k].cpu().reshape(-1)
    if key + "full" in f.files:
        ref, got = f[key + "full"], t.numpy()
    else:
        ref, got = f[key + "val"], t[torch.from_numpy(f[key + "idx"])].numpy()
    print("%-45s max|d| %.2e  ref max %.2e  rel %.2e" % (k, np.abs(got - ref).max(), np.abs(ref).max(), np.abs(got - ref).max() / np.abs(ref).max()))
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(batch["image"].to(DEV), torch.from_numpy(f["normal"]).to(DEV), din[:, None].to(DEV), gt.to(DEV))
    torch.cuda.synchronize(); print("step wall %.1f ms" % (1e3 * (time.perf_counter() - t0)))
