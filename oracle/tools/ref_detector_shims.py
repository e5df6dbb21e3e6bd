"""Import the reference's plane-mask detector (plane_mask_detection/, a maskrcnn_benchmark fork) on CPU, in the build
container only.  TEST INFRASTRUCTURE: used by `oracle/tools/make_golden_detector.py` to produce golden vectors; nothing from the
reference is copied into this repo and nothing here runs on the GPU box.

The fork needs packages this image lacks; each gets the smallest stand-in that lets the *reference's own Python* run:

  * yacs.config.CfgNode        -> `CfgNode` below: attribute dict with merge_from_file (PyYAML) / merge_from_list / freeze / clone
  * apex.amp.float_function    -> identity decorator
  * maskrcnn_benchmark._C      -> `nms`, `roi_align_forward` bound to this repo's oracle restatements (oracle/detector_oracle.py,
                                  NMS pinned to the reference's own test vectors); the reference's C++ does not compile against this
                                  PyTorch, so there is no native build to bind instead
  * cv2, pycocotools, matplotlib, torchvision -> empty modules (only touched by code paths the golden script does not call), plus
                                  torchvision.transforms with the few classes demo/predictor.py composes
"""
import copy
import sys
import types

import numpy as np
import torch
import yaml

REFERENCE_ROOT = "/root/reference"


class CfgNode(dict):
    """Minimal yacs.config.CfgNode: nested attribute dict."""

    def __init__(self, init=None, **kw):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def freeze(self):
        pass

    def defrost(self):
        pass

    def _merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), CfgNode):
                self[k]._merge(v)
            else:
                old = self.get(k)
                if isinstance(old, tuple) and isinstance(v, list):
                    v = tuple(v)
                if isinstance(v, str) and isinstance(old, tuple):
                    v = tuple(eval(v))
                self[k] = v

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f))

    def merge_from_list(self, lst):
        for k, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = v


def install():
    if getattr(install, "_done", False):
        return
    install._done = True
    from oracle import detector_oracle as DO

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("yacs")
    mod("yacs.config", CfgNode=CfgNode)
    mod("apex", amp=types.SimpleNamespace(float_function=lambda f: f, half_function=lambda f: f))
    sys.modules["apex.amp"] = sys.modules["apex"].amp

    def nms(dets, scores, thr):
        return torch.from_numpy(DO.nms(dets.numpy(), scores.numpy(), float(thr)))

    def roi_align_forward(inp, rois, scale, ph, pw, ratio):
        return torch.from_numpy(DO.roi_align_forward(inp.numpy(), rois.numpy(), float(scale), int(ph), int(pw), int(ratio)))

    cmod = types.SimpleNamespace(nms=nms, roi_align_forward=roi_align_forward)
    mod("maskrcnn_benchmark", _C=cmod)
    mod("cv2")
    mod("pycocotools")
    mod("pycocotools.mask")
    mod("matplotlib")
    mod("matplotlib.pyplot")

    class _T:       # torchvision.transforms: only composed, never called by the golden script
        class Compose:
            def __init__(self, ts):
                self.transforms = ts

        class Lambda:
            def __init__(self, f):
                self.f = f

        class Normalize:
            def __init__(self, mean, std):
                self.mean, self.std = mean, std

        class ToPILImage:
            pass

        class ToTensor:
            pass

    tvt = mod("torchvision.transforms", Compose=_T.Compose, Lambda=_T.Lambda, Normalize=_T.Normalize, ToPILImage=_T.ToPILImage,
              ToTensor=_T.ToTensor)
    mod("torchvision.transforms.functional")
    tvt.functional = sys.modules["torchvision.transforms.functional"]
    mod("torchvision", transforms=tvt)
    np.float = float
    if not hasattr(torch, "_six"):       # removed from modern torch; utils/imports.py:4 tests torch._six.PY3
        torch._six = types.SimpleNamespace(PY3=True, string_classes=(str,))
    import torch.hub as _hub
    if not hasattr(_hub, "_download_url_to_file"):      # utils/model_zoo.py:6 (only imported, never called: no network)
        _hub._download_url_to_file = None
        _hub.urlparse = getattr(_hub, "urlparse", None)
        _hub.HASH_REGEX = getattr(_hub, "HASH_REGEX", None)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def build_reference_detector(config="configs/R101_bs16_all_plane_normal.yaml"):
    """The reference's GeneralizedRCNN for its shipped config, on CPU, eval mode, default-initialised."""
    install()
    from plane_mask_detection.maskrcnn_benchmark.config import cfg
    from plane_mask_detection.maskrcnn_benchmark.modeling.detector import build_detection_model
    cfg = cfg.clone()
    cfg.merge_from_file(REFERENCE_ROOT + "/plane_mask_detection/" + config)
    cfg.merge_from_list(["MODEL.DEVICE", "cpu"])
    model = build_detection_model(cfg)
    model.eval()
    return model, cfg


if __name__ == "__main__":
    sys.path.insert(0, "/root/repo")
    m, cfg = build_reference_detector()
    sd = m.state_dict()
    print(len(sd), "state_dict entries;", sum(v.numel() for v in sd.values()) / 1e6, "M values")
    for k in list(sd)[:8]:
        print(k, tuple(sd[k].shape))
