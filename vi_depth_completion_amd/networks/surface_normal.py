"""Drop-in for the reference's `SurfaceNormalPrediction` (networks/surface_normal.py:57-171): same constructor,
`forward(x, gravity_tensor, alignment_tensor)` and the same 759 state_dict keys, executed as one HIP program:
warp -> ResNet-101 pyramid -> 4-branch decoder -> head -> inverse warp + R^T + L2-normalise.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import engine
from .backbone import ResNetPyramids
from .fpn_decoder import build_branch, emit_decoder
from .warping_2dof_alignment import Warping2DOFAlignment


class _HipModule(nn.Module):
    """Shared plumbing: derived-weight cache invalidation and per-batch-size program cache."""

    def _init_engine(self):
        self._weights = engine.WeightStore(self)
        self._programs = {}
        self._version = 0          # bumped whenever the parameters may have changed (programs spanning modules key on it)
        self.register_load_state_dict_post_hook(lambda m, keys: m._invalidate())

    def _invalidate(self):
        self._weights.invalidate()
        self._programs.clear()
        self._version += 1

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        if hasattr(self, "_weights"):
            self._invalidate()
        return r

    def _replicate_for_data_parallel(self):
        # torch.nn.DataParallel (network_run.py:97-99) copies a module per GPU and re-broadcasts its parameters on every forward; a
        # replica of this module would share the packed weights and captured graphs of GPU 0.  Refused, not emulated: frames shard one
        # process per GPU (vi_depth_completion_amd/sharding.py, bench.py --gpus N).
        raise RuntimeError("%s: torch.nn.DataParallel is not supported by the HIP path; run one process per GPU "
                           "(vi_depth_completion_amd.sharding, bench.py --gpus N)" % type(self).__name__)

    def _check(self, *tensors):
        if self.training:
            raise RuntimeError("%s: the HIP path is inference-only (BatchNorm is folded); call .eval()" % type(self).__name__)
        for t in tensors:
            if not t.is_cuda:
                raise RuntimeError("%s runs on the GPU only: there is no CPU/eager fallback" % type(self).__name__)

    @staticmethod
    def _execute(prog):
        mode = os.environ.get("VIDC_EXEC", "graph")
        if mode == "graph":
            if not prog.captured:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    prog.run()            # warm-up outside capture (sets kernel attributes)
                    prog.capture()
                torch.cuda.current_stream().wait_stream(side)
            prog.launch()
        else:
            prog.run()


class SurfaceNormalPrediction(_HipModule):
    def __init__(self, output_size=(240, 320), in_channels=3, training_mode="train_L2_loss",
                 fc_img=np.array([0.5 * 577.87061, 0.5 * 580.25851]),
                 cc_img=np.array([0.5 * 319.87654, 0.5 * 239.87603]), use_mask=False, align_corners=False):
        super().__init__()
        self.output_size, self.mode, self.use_mask = output_size, training_mode, use_mask
        self.warp_2dof_alignment = Warping2DOFAlignment(fx=fc_img[0], fy=fc_img[1], cx=cc_img[0], cy=cc_img[1],
                                                        align_corners=align_corners)
        self.resnet_pyramids = ResNetPyramids(in_channels=in_channels)
        for lvl in (1, 2, 3, 4):
            setattr(self, "feature%d_upsamping" % lvl, build_branch(lvl, 1))
        self.feature_concat = nn.Sequential(nn.Conv2d(128, 64, 3, 1, 1), nn.ReLU(inplace=True), nn.Conv2d(64, 3, 1),
                                            nn.UpsamplingBilinear2d(size=output_size))
        self._init_engine()

    def build_program(self, B, device, dry_run=False):
        wp = self.warp_2dof_alignment
        prog = engine.Program(self._weights, device, B)
        x = prog.input_nchw("image", self.resnet_pyramids.channel, wp.H, wp.W)
        g = prog.input_raw("gravity", B * 3)
        a = prog.input_raw("aligned", B * 3)
        kinv = prog.input_raw("kinv", 9)
        params = prog.warp_params(g, a, wp, kinv)
        xw = prog.warp_fwd(x, params, wp, wp.align_corners)
        levels = self.resnet_pyramids.emit(prog, xw, engine.K("resnet_pyramids."))
        if self.use_mask:               # surface_normal.py:150-162: features and their sum are zeroed where the warp left no image
            zsum = prog.mask_scale(emit_decoder(prog, self, [prog.mask_scale(t, xw) for t in levels]), xw)
        else:
            zsum = emit_decoder(prog, self, levels)
        h = prog.conv(zsum, "feature_concat.0", relu=True, padding=1)
        y, low = prog.head(h, "feature_concat.2", 0, (wp.H, wp.W), relu=False)
        z = prog.warp_inv(y, params, wp, wp.align_corners, normalize=True)
        prog.mark_output("normals", z)
        prog.mark_output("warped", xw)
        prog.mark_output("warp_params", params)
        prog.taps = {"x%d" % (i + 1): t for i, t in enumerate(levels)}
        prog.taps.update(zsum=zsum, normal_raw=y)
        prog.finalize(dry_run)
        prog.storage[kinv.buf][:9].copy_(wp.kinv(device))
        return prog

    def program(self, B, device, slot=0):
        """Programs are cached per batch size and per `slot` (independent activation buffers sharing the weights), so
        several frames can be in flight on different HIP streams."""
        key = (B, str(device), self.warp_2dof_alignment.align_corners, slot)
        if key not in self._programs:
            self._programs[key] = self.build_program(B, device)
        return self._programs[key]

    def enqueue(self, x, gravity_tensor, alignment_tensor, slot=0):
        """Enqueues one forward pass on the current stream; returns a VIEW of the program's output buffer (valid
        until the next enqueue on the same slot)."""
        self._check(x, gravity_tensor, alignment_tensor)
        B = x.shape[0]
        prog = self.program(B, x.device, slot)
        prog.tensor(prog.inputs["image"]).copy_(x, non_blocking=True)
        prog.storage[prog.inputs["gravity"].buf][: B * 3].copy_(gravity_tensor.reshape(-1), non_blocking=True)
        prog.storage[prog.inputs["aligned"].buf][: B * 3].copy_(alignment_tensor.reshape(-1), non_blocking=True)
        self._execute(prog)
        return prog.tensor(prog.outputs["normals"])

    def forward(self, x, gravity_tensor, alignment_tensor):
        return self.enqueue(x, gravity_tensor, alignment_tensor).clone()
