"""Drop-in for the reference's `ModifiedFPN` (networks/depth_completion.py:68-165): no-argument constructor,
`forward(image, normal, incomplete_depth)` and the same 2031 state_dict keys.  The three ResNet-101 pyramids run
as ONE grouped launch per layer (groups = rgb, normal, depth) writing channel slices of a shared buffer, so
`combine_rgbdn` (torch.cat, :151-152) is free; the 3x-wide decoder follows.
"""
import torch
import torch.nn as nn

from .. import engine
from .backbone import ResNetPyramids
from .fpn_decoder import build_branch, emit_decoder
from .surface_normal import _HipModule


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


class ModifiedFPN(_HipModule):
    def __init__(self, resnet_arch=101):
        super().__init__()
        self.resnet_rgb = ResNetPyramids(in_channels=3, pretrained=True, resnet_arch=resnet_arch)
        self.resnet_normal = ResNetPyramids(in_channels=3, pretrained=False, resnet_arch=resnet_arch)
        self.resnet_depth = ResNetPyramids(in_channels=1, pretrained=False, resnet_arch=resnet_arch)
        for lvl in (1, 2, 3, 4):
            setattr(self, "feature%d_upsamping" % lvl, build_branch(lvl, 3))
        # NB: the reference pads its last 1x1 conv (Conv2d(192, 1, 1, 1, 1)) -> 62x82 map; reproduced by the head kernel
        self.feature_concat = nn.Sequential(nn.Conv2d(128 * 3, 64 * 3, 3, 1, 1), nn.ReLU(inplace=True),
                                            nn.Conv2d(64 * 3, 1, 1, 1, 1), nn.UpsamplingBilinear2d(size=(240, 320)),
                                            nn.ReLU(inplace=True))
        self._init_engine()

    def combine_rgbdn(self, rgb, normal, depth, level):
        return torch.cat((rgb, normal, depth), dim=1)

    def build_program(self, B, H, W, device, dry_run=False):
        prog = engine.Program(self._weights, device, B)
        img = prog.input_nchw("image", 3, H, W)
        nrm = prog.input_nchw("normal", 3, H, W)
        dep = prog.input_nchw("depth", 1, H, W)
        levels = self.resnet_rgb.emit(prog, [img, nrm, dep], engine.K(("resnet_rgb.", "resnet_normal.", "resnet_depth.")))
        zsum = emit_decoder(prog, self, levels)
        h = prog.conv(zsum, "feature_concat.0", relu=True, padding=1)
        y, low = prog.head(h, "feature_concat.2", 1, (H, W), relu=True)
        prog.mark_output("depth", y)
        prog.mark_output("head_lowres", low)
        prog.taps = {"x%d" % (i + 1): t for i, t in enumerate(levels)}
        prog.taps.update(zsum=zsum)
        prog.finalize(dry_run)
        return prog

    def program(self, B, H, W, device, slot=0):
        key = (B, H, W, str(device), slot)
        if key not in self._programs:
            self._programs[key] = self.build_program(B, H, W, device)
        return self._programs[key]

    def enqueue(self, image, normal, incomplete_depth, slot=0):
        """Enqueues one forward pass on the current stream; returns a VIEW of the program's output buffer."""
        self._check(image, normal, incomplete_depth)
        B, _, H, W = image.shape
        prog = self.program(B, H, W, image.device, slot)
        prog.tensor(prog.inputs["image"]).copy_(image, non_blocking=True)
        prog.tensor(prog.inputs["normal"]).copy_(normal, non_blocking=True)
        prog.tensor(prog.inputs["depth"]).copy_(incomplete_depth, non_blocking=True)
        self._execute(prog)
        return prog.tensor(prog.outputs["depth"])

    def forward(self, image, normal, incomplete_depth):
        return self.enqueue(image, normal, incomplete_depth).clone()
