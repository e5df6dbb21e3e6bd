"""Import the *reference itself* (/root/reference) on CPU, in the build container only.

TEST INFRASTRUCTURE.  Used by `oracle/tools/make_golden.py` to generate golden vectors; never runs on
the GPU box (the reference is not there) and nothing from the reference is copied into this repo.
The obstacles and their shims are the ones listed in SURVEY.md §8c:

  * hard-wired 'cuda:0' / .cuda() / torch.cuda.FloatTensor     -> identity on CPU
  * torchvision (absent): resnet101 topology + transforms.ToTensor -> stubs defined below
  * cv2 / skimage / yacs / apex / maskrcnn `_C` (absent)       -> empty stub modules; COCODemo returns a fixed id map
  * np.float (removed in numpy >= 1.24)                        -> float
  * `mask[mask] = values` with the index aliasing the target (main.py:157) -> index cloned
"""
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


# ---- torchvision stub: Bottleneck ResNet with torchvision's attribute names -------------------
class _Bottleneck(nn.Module):
    def __init__(self, cin, planes, stride, project):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        t = self.relu(self.bn1(self.conv1(x)))
        t = self.relu(self.bn2(self.conv2(t)))
        t = self.bn3(self.conv3(t))
        return self.relu(t + idn)


class _ResNet(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), blocks)):
            stride = 1 if i == 0 else 2
            mods = [_Bottleneck(cin, planes, stride, True)] + [_Bottleneck(planes * 4, planes, 1, False) for _ in range(n - 1)]
            setattr(self, "layer%d" % (i + 1), nn.Sequential(*mods))
            cin = planes * 4


def _resnet101(pretrained=False, **kw):
    return _ResNet((3, 4, 23, 3))


def _resnet50(pretrained=False, **kw):
    return _ResNet((3, 4, 6, 3))


class _ToTensor:
    def __call__(self, pic):
        arr = np.asarray(pic)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
        return t.float().div(255) if t.dtype == torch.uint8 else t.float()


class FixedPlaneMask:
    """Stand-in for COCODemo (plane_mask_detection/demo/predictor.py:143-150): returns a fixed id map."""
    id_map = None

    def __init__(self, *a, **k):
        pass

    def run_on_tensor(self, image):
        return FixedPlaneMask.id_map.copy()


def install():
    """Idempotently install all shims and put the reference on sys.path."""
    if getattr(install, "_done", False):
        return
    install._done = True

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    tv_models = mod("torchvision.models", resnet101=_resnet101, resnet50=_resnet50)
    tv_tf = mod("torchvision.transforms.functional", to_tensor=_ToTensor(), to_pil_image=None)
    tv_t = mod("torchvision.transforms", ToTensor=_ToTensor, functional=tv_tf)
    mod("torchvision", models=tv_models, transforms=tv_t)
    mod("skimage.io")
    mod("skimage", io=sys.modules["skimage.io"])
    mod("cv2")
    cfg = types.SimpleNamespace(merge_from_file=lambda f: None)
    mod("plane_mask_detection")
    mod("plane_mask_detection.maskrcnn_benchmark")
    mod("plane_mask_detection.maskrcnn_benchmark.config", cfg=cfg)
    mod("plane_mask_detection.demo")
    mod("plane_mask_detection.demo.predictor", COCODemo=FixedPlaneMask)

    np.float = float
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self

    _to = torch.Tensor.to

    def to(self, *a, **k):
        if a and isinstance(a[0], str) and a[0].startswith("cuda"):
            return self
        return _to(self, *a, **k)

    torch.Tensor.to = to

    _setitem = torch.Tensor.__setitem__

    def setitem(self, idx, val):
        if idx is self:
            idx = idx.clone()
        return _setitem(self, idx, val)

    torch.Tensor.__setitem__ = setitem
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def load_reference():
    """Returns the reference's `main` module (with networks/, dataset, network_run imported)."""
    install()
    import main as ref_main  # noqa: the reference's main.py
    return ref_main
