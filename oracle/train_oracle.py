"""CPU restatement of the reference's training iteration for the depth-completion network: TEST INFRASTRUCTURE (imported only by
tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke()).

Follows `ImageNetworkRunInterface._run_training_iteration` (network_run.py:231-254): `cnn.train()` (BatchNorm on batch statistics,
running statistics updated), forward, `_network_loss` (network_run.py:158-191: `L1Loss(reduction='sum')` over `depth > 0`, divided by
H*W of the image), `backward()`, `torch.optim.Adam(cnn.parameters(), lr)` step (network_run.py:228-229: defaults betas (0.9, 0.999),
eps 1e-8, no weight decay).  The network inputs (image, predicted normals, enriched sparse depth) are what `_call_cnn` (main.py:261-298)
hands to `self.cnn`; they do not depend on the trained parameters, so the step is a function of (parameters, inputs, ground truth).

PINNED against the reference itself: oracle/tools/make_golden_train.py imports network_run.py / main.py, runs one
`_run_training_iteration` on a 2-frame batch and stores loss, gradient and updated-parameter probes (tests/golden/train_step.npz)."""
import contextlib
import math

import torch

from . import vidc_oracle as O


@contextlib.contextmanager
def bn_training():
    old = O.BN_TRAINING
    O.BN_TRAINING = True
    try:
        yield
    finally:
        O.BN_TRAINING = old


def is_parameter(name):
    return not (name.endswith("running_mean") or name.endswith("running_var") or name.endswith("num_batches_tracked"))


def depth_l1_loss(pred, gt):
    """network_run.py:163-173: sum |pred - gt| over gt > 0, divided by the image size H*W (not by the number of valid pixels,
    not by the batch size)."""
    _, _, H, W = pred.shape
    mask = gt > 0
    return torch.nn.functional.l1_loss(pred[mask], gt[mask], reduction="sum") / (H * W)


def forward_backward(sd, image, normal, depth_in, gt):
    """One train-mode forward + backward.  `sd`: state_dict-like {name: tensor}; parameters are cloned as leaves, buffers (running
    statistics) are cloned and updated like nn.BatchNorm2d does.  Returns (loss, pred, {name: grad}, {name: updated buffer})."""
    work = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            work[k] = v.clone()
        elif is_parameter(k):
            work[k] = v.detach().clone().requires_grad_(True)
        else:
            work[k] = v.detach().clone()
    with torch.enable_grad(), bn_training():
        pred = O.depth_completion_forward(work, image, normal, depth_in)
        loss = depth_l1_loss(pred, gt)
        loss.backward()
    grads = {k: v.grad for k, v in work.items() if is_parameter(k) and torch.is_tensor(v) and v.requires_grad}
    bufs = {k: v for k, v in work.items() if not is_parameter(k) and not k.endswith("num_batches_tracked")}
    return loss.detach(), pred.detach(), grads, bufs


def adam_step(params, grads, state, lr, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad), restated: state = {"step": int, "m": {...}, "v": {...}}; returns new params."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
    out = {}
    for k, p in params.items():
        g = grads[k]
        m = state.setdefault("m", {}).get(k)
        v = state.setdefault("v", {}).get(k)
        m = torch.zeros_like(p) if m is None else m
        v = torch.zeros_like(p) if v is None else v
        m = m * b1 + g * (1 - b1)
        v = v * b2 + g * g * (1 - b2)
        state["m"][k], state["v"][k] = m, v
        denom = v.sqrt() / math.sqrt(bc2) + eps
        out[k] = p - (lr / bc1) * (m / denom)
    return out


def training_iteration(sd, image, normal, depth_in, gt, lr, state):
    """network_run.py:231-254 for the depth-completion network.  Returns (loss, new state_dict)."""
    loss, _pred, grads, bufs = forward_backward(sd, image, normal, depth_in, gt)
    params = {k: v for k, v in sd.items() if is_parameter(k)}
    new = adam_step(params, grads, state, lr)
    out = dict(sd)
    out.update(new)
    out.update(bufs)
    for k in sd:
        if k.endswith("num_batches_tracked"):
            out[k] = sd[k] + 1
    return loss, out
