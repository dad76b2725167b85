"""SimpleBaseline (ResNet backbone + transposed-conv head) on the HIP engine.

Drop-in for the reference's ``src/modeling/simplebaseline/pose_resnet.py``: same factory
(``get_pose_net(cfg, is_train)``), same attribute names and ``state_dict`` keys/layouts
(conv1, bn1, layer1..4.<b>.{conv,bn}{1,2,3}, layerL.0.downsample.{0,1}, deconv_layers.{0..8},
final_layer), same default initialisation order (so ``torch.manual_seed(s)`` reproduces the
reference's random weights bit for bit) -- but ``forward`` is executed by
``lighthand_amd.engine`` with hand-written gfx950 kernels.  The torch.nn layers created here
only own parameters/buffers; their own ``forward`` is never called.
"""
import torch.nn as nn

from ...module import HipModule

BN_MOMENTUM = 0.1          # reference pose_resnet.py:19

# depth -> (unit kind, units per stage); reference pose_resnet.py:301-305
resnet_spec = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottleneck", [3, 4, 6, 3]),
               101: ("bottleneck", [3, 4, 23, 3]), 152: ("bottleneck", [3, 8, 36, 3])}
_EXPANSION = {"basic": 1, "bottleneck": 4}


class ResidualUnit(nn.Module):
    """Parameter holder for one BasicBlock / Bottleneck (reference pose_resnet.py:29-141).

    ``stride_on`` says which convolution carries the stride: the 3x3 ('pytorch' style) or the
    first 1x1 ('caffe' style, reference :102-141)."""

    def __init__(self, kind, inplanes, planes, stride, downsample, caffe=False):
        super().__init__()
        self.kind, self.stride, self.caffe = kind, stride, caffe
        if kind == "basic":
            self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
            self.relu = nn.ReLU(inplace=True)
            self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        else:
            s1, s2 = (stride, 1) if caffe else (1, stride)
            self.conv1 = nn.Conv2d(inplanes, planes, 1, s1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
            self.conv2 = nn.Conv2d(planes, planes, 3, s2, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
            self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
            self.bn3 = nn.BatchNorm2d(planes * 4, momentum=BN_MOMENTUM)
            self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def describe(self, gb, prefix, x):
        """Emit this unit's nodes; returns the output activation.  The projection shortcut is described FIRST: in the
        reversed (backward) order its data gradient then comes after conv1's, which is dense, so conv1 stores the block
        input's gradient outright and the shortcut's stride-2 gradient only adds its one non-zero sub-pixel phase."""
        r = gb.conv(x, prefix + ".downsample.0", 1, self.stride, 0) if self.downsample is not None else None
        if self.kind == "basic":
            y = gb.conv(x, prefix + ".conv1", 3, self.stride, 1)
            a = gb.fuse([(y, prefix + ".bn1")])
            y = gb.conv(a, prefix + ".conv2", 3, 1, 1)
            last = prefix + ".bn2"
        else:
            s1, s2 = (self.stride, 1) if self.caffe else (1, self.stride)
            y = gb.conv(x, prefix + ".conv1", 1, s1, 0)
            a = gb.fuse([(y, prefix + ".bn1")])
            y = gb.conv(a, prefix + ".conv2", 3, s2, 1)
            a = gb.fuse([(y, prefix + ".bn2")])
            y = gb.conv(a, prefix + ".conv3", 1, 1, 0)
            last = prefix + ".bn3"
        if self.downsample is not None:
            return gb.fuse([(y, last), (r, prefix + ".downsample.1")])
        return gb.fuse([(y, last), x])


def make_stage(kind, inplanes, planes, units, stride=1, caffe=False):
    """Reference ``_make_layer`` (pose_resnet.py:177-192): the projection shortcut is created
    BEFORE the first unit (this fixes the random-init order)."""
    exp = _EXPANSION[kind]
    downsample = None
    if stride != 1 or inplanes != planes * exp:
        downsample = nn.Sequential(nn.Conv2d(inplanes, planes * exp, 1, stride, bias=False),
                                   nn.BatchNorm2d(planes * exp, momentum=BN_MOMENTUM))
    seq = [ResidualUnit(kind, inplanes, planes, stride, downsample, caffe)]
    seq += [ResidualUnit(kind, planes * exp, planes, 1, None, caffe) for _ in range(1, units)]
    return nn.Sequential(*seq), planes * exp


def describe_stage(gb, seq, prefix, x):
    for i, unit in enumerate(seq):
        x = unit.describe(gb, f"{prefix}.{i}", x)
    return x


class PoseResNet(HipModule):

    def __init__(self, kind, units, cfg, caffe=False, **kwargs):
        super().__init__()
        extra = cfg.MODEL.EXTRA
        self.deconv_with_bias = extra.DECONV_WITH_BIAS
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        c = 64
        self.layer1, c = make_stage(kind, c, 64, units[0], 1, caffe)
        self.layer2, c = make_stage(kind, c, 128, units[1], 2, caffe)
        self.layer3, c = make_stage(kind, c, 256, units[2], 2, caffe)
        self.layer4, c = make_stage(kind, c, 512, units[3], 2, caffe)
        assert extra.NUM_DECONV_LAYERS == len(extra.NUM_DECONV_FILTERS) == len(extra.NUM_DECONV_KERNELS), \
            "ERROR: num_deconv_layers is different len(num_deconv_filters)"
        head = []
        for planes, k in zip(extra.NUM_DECONV_FILTERS, extra.NUM_DECONV_KERNELS):
            pad, opad = {4: (1, 0), 3: (1, 1), 2: (0, 0)}[k]      # reference :194-205
            head += [nn.ConvTranspose2d(c, planes, k, 2, pad, opad, bias=self.deconv_with_bias),
                     nn.BatchNorm2d(planes, momentum=BN_MOMENTUM), nn.ReLU(inplace=True)]
            c = planes
        self.deconv_layers = nn.Sequential(*head)
        fk = extra.FINAL_CONV_KERNEL
        # 21 output joints are hard-wired in the reference (:171)
        self.final_layer = nn.Conv2d(c, 21, fk, 1, 1 if fk == 3 else 0)

    def describe(self, gb):
        x = gb.input()
        x = gb.fuse([(gb.conv(x, "conv1", 7, 2, 3), "bn1")])
        x = gb.maxpool(x)
        for name in ("layer1", "layer2", "layer3", "layer4"):
            x = describe_stage(gb, getattr(self, name), name, x)
        for i in range(0, len(self.deconv_layers), 3):
            dc = self.deconv_layers[i]
            y = gb.deconv(x, f"deconv_layers.{i}", dc.kernel_size[0],
                          bias=f"deconv_layers.{i}.bias" if dc.bias is not None else None)
            x = gb.fuse([(y, f"deconv_layers.{i + 1}")])
        fk = self.final_layer.kernel_size[0]
        gb.output(gb.conv(x, "final_layer", fk, 1, 1 if fk == 3 else 0, bias="final_layer.bias"))

    def init_weights(self, pretrained=""):
        """The reference never calls this (both factories have it commented out);
        kept for API parity: ImageNet checkpoints are loaded with strict=False like :250-298."""
        import os
        import torch
        if not os.path.isfile(pretrained):
            raise ValueError("imagenet pretrained model does not exist")
        for m in self.deconv_layers.modules():
            if isinstance(m, nn.ConvTranspose2d):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.final_layer.weight, std=0.001)
        nn.init.constant_(self.final_layer.bias, 0)
        ckpt = torch.load(pretrained, map_location="cpu")
        sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        self.load_state_dict(sd, strict=False)


def get_pose_net(cfg, is_train, **kwargs):
    """Factory with the reference's signature (pose_resnet.py:308-322)."""
    kind, units = resnet_spec[cfg.MODEL.EXTRA.NUM_LAYERS]
    return PoseResNet(kind, units, cfg, caffe=(cfg.MODEL.STYLE == "caffe"), **kwargs)
