"""Drop-in for the reference's `SurfaceNormalDORN` (networks/surface_normal_dorn.py:143-154), the surface-normal network of the
`--use_gravity 0` branch (main.py:244-245): same constructor, `forward(x)` and the same state_dict keys, executed as one HIP program:
ResNet-101 with the strides of layer3/layer4 removed (:119-125, features at 1/8 resolution) -> scene-understanding module (global
encoder: AvgPool2d(8,8,(1,0)) -> Linear(40960,512) -> 1x1 conv -> broadcast; ASPP: 1x1 and three dilated 3x3 branches, dilation
6/12/18; concat -> 1x1 -> 1x1 -> bilinear to the output size) -> F.normalize.  Dropout2d layers are identities in eval mode.
"""
import collections

import torch
import torch.nn as nn

from .. import engine
from .backbone import _conv, _stage, STAGE_BLOCKS, STAGE_PLANES
from .surface_normal import _HipModule


class FullImageEncoder(nn.Module):
    def __init__(self, dataset="kitti"):
        super().__init__()
        self.global_pooling = nn.AvgPool2d(8, stride=8, padding=(1, 0))
        self.dropout = nn.Dropout2d(p=0.5)
        self.global_fc = nn.Linear(2048 * 4 * 5, 512)
        self.relu = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(512, 512, 1)
        self.upsample = nn.UpsamplingBilinear2d(size=(30, 40))
        self.dataset = dataset


def _aspp(dilation):
    if dilation == 0:
        first = nn.Conv2d(2048, 512, 1)
    else:
        first = nn.Conv2d(2048, 512, 3, padding=dilation, dilation=dilation)
    return nn.Sequential(first, nn.BatchNorm2d(512), nn.ReLU(inplace=True), nn.Conv2d(512, 512, 1), nn.BatchNorm2d(512), nn.ReLU(inplace=True))


class SceneUnderstandingModuleBN(nn.Module):
    def __init__(self, output_channel=136, dataset="kitti", mode="L2"):
        super().__init__()
        self.encoder = FullImageEncoder(dataset=dataset)
        self.aspp1, self.aspp2, self.aspp3, self.aspp4 = _aspp(0), _aspp(6), _aspp(12), _aspp(18)
        self.concat_process = nn.Sequential(nn.Dropout2d(p=0.5), nn.Conv2d(512 * 5, 2048, 1), nn.ReLU(inplace=True), nn.Dropout2d(p=0.5),
                                            nn.Conv2d(2048, output_channel, 1), nn.UpsamplingBilinear2d(size=(240, 320)))


class ResNet(nn.Module):
    """`feature_extractor`: the stem of ResNetPyramids, torchvision's layer1..4 with layer3[0] / layer4[0] at stride 1."""

    def __init__(self, in_channels=3, pretrained=True):
        super().__init__()
        del pretrained
        self.channel = in_channels
        self.conv1 = nn.Sequential(collections.OrderedDict([
            ("conv1_1", _conv(in_channels, 64, 3, 2, 1)), ("relu1_1", nn.ReLU(inplace=True)),
            ("conv1_2", _conv(64, 64, 3, 1, 1)), ("bn_2", nn.BatchNorm2d(64)), ("relu1_2", nn.ReLU(inplace=True)),
            ("conv1_3", _conv(64, 128, 3, 1, 1)), ("bn1_3", nn.BatchNorm2d(128)), ("relu1_3", nn.ReLU(inplace=True))]))
        self.bn1 = nn.BatchNorm2d(128)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        cin = 128
        for i, (planes, n, stride) in enumerate(zip(STAGE_PLANES, STAGE_BLOCKS[101], (1, 2, 1, 1))):
            setattr(self, "layer%d" % (i + 1), _stage(cin, planes, n, stride))
            cin = planes * 4

    def emit(self, prog, x, prefix):
        p = prefix + "conv1."
        t = prog.stem_conv(x, p + "conv1_1", relu=True)
        t = prog.conv(t, p + "conv1_2", bn=p + "bn_2", relu=True, padding=1)
        t = prog.conv(t, p + "conv1_3", bn=p + "bn1_3", relu=True, padding=1, bn2=prefix + "bn1", relu2=True)
        t = prog.maxpool(t)
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(self, "layer%d" % li)):
                t = blk.emit(prog, t, prefix + "layer%d.%d." % (li, bi))
        return t


class SurfaceNormalDORN(_HipModule):
    def __init__(self, output_size=(240, 320), pretrained=True, output_channel=3, training_mode="train_L2_loss"):
        super().__init__()
        self.output_size = output_size
        self.feature_extractor = ResNet(pretrained=pretrained)
        self.aspp_module = SceneUnderstandingModuleBN(output_channel=output_channel, mode=training_mode)
        self._init_engine()

    def build_program(self, B, H, W, device, dry_run=False):
        prog = engine.Program(self._weights, device, B)
        x = prog.input_nchw("image", 3, H, W)
        f = self.feature_extractor.emit(prog, x, engine.K("feature_extractor."))          # (B, H/8, W/8, 2048)
        a = "aspp_module."
        cat = prog.nhwc(f.H, f.W, 512 * 5)                                                 # torch.cat((x1..x5), dim=1): channel slices
        sl = lambda k: engine.T(cat.buf, cat.B, cat.H, cat.W, 512, 1, cat.ld, 512 * k)
        # x1: full-image encoder (:18-30); Dropout2d is the identity in eval mode
        e = prog.avgpool(f, (8, 8), (8, 8), (1, 0))
        e = prog.linear(e, a + "encoder.global_fc", relu=True)
        e = prog.conv(e, a + "encoder.conv1")
        prog.upsample(e, (f.H, f.W), out=sl(0))                                            # from 1x1: a broadcast
        # x2..x5: ASPP (:37-68)
        for k, (name, dil) in enumerate((("aspp1", 0), ("aspp2", 6), ("aspp3", 12), ("aspp4", 18)), start=1):
            if dil == 0:
                t = prog.conv(f, a + name + ".0", bn=a + name + ".1", relu=True)
            else:
                t = prog.conv(f, a + name + ".0", bn=a + name + ".1", relu=True, padding=dil, dilation=dil)
            prog.conv(t, a + name + ".3", bn=a + name + ".4", relu=True, out=sl(k))
        h = prog.conv(cat, a + "concat_process.1", relu=True)
        y, _low = prog.head(h, a + "concat_process.4", 0, (H, W), relu=False)              # 1x1 to 3 channels + UpsamplingBilinear2d
        z = prog.normalize_nchw(y)
        prog.mark_output("normals", z)
        prog.taps = {"features": f, "concat": cat}
        prog.finalize(dry_run)
        return prog

    def program(self, B, H, W, device):
        key = (B, H, W, str(device))
        if key not in self._programs:
            self._programs[key] = self.build_program(B, H, W, device)
        return self._programs[key]

    def forward(self, x):
        self._check(x)
        B, _, H, W = x.shape
        if (H, W) != tuple(self.output_size):
            raise ValueError("SurfaceNormalDORN was built for %s inputs, got %s" % (tuple(self.output_size), (H, W)))
        prog = self.program(B, H, W, x.device)
        prog.tensor(prog.inputs["image"]).copy_(x, non_blocking=True)
        self._execute(prog)
        return prog.tensor(prog.outputs["normals"]).clone()
