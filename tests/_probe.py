"""Shared by the training tests: a tensor against its stored summary in tests/golden/train_step.npz."""
import numpy as np
import torch


def check_probe(f, tag, name, t, rtol, atol):
    """A tensor against its stored summary (whole tensor if small, else sum / |sum| / 64 probe elements)."""
    t = t.detach().float().reshape(-1)
    key = "%s|%s|" % (tag, name)
    if key + "full" in f.files:
        ref = f[key + "full"]
        assert np.abs(t.numpy() - ref).max() <= atol + rtol * np.abs(ref).max(), (tag, name, np.abs(t.numpy() - ref).max(), np.abs(ref).max())
    else:
        ref = f[key + "val"]
        got = t[torch.from_numpy(f[key + "idx"])].numpy()
        scale = float(f[key + "abs"]) / t.numel()
        assert np.abs(got - ref).max() <= atol + rtol * max(np.abs(ref).max(), scale), (tag, name, np.abs(got - ref).max(), scale)
        assert abs(float(t.double().abs().sum()) - float(f[key + "abs"])) <= (atol * t.numel() + rtol * float(f[key + "abs"])), (tag, name)
