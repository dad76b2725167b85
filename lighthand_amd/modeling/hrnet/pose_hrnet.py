"""HRNet (any width: W32 / W48 / ...) on the HIP engine.

Drop-in for the reference's ``src/modeling/hrnet/pose_hrnet.py``: ``get_hrnet(cfg_dict,
is_train)`` takes the dict ``yaml.safe_load`` gives for ``config/cfg.yaml``; attribute names
and ``state_dict`` keys (conv1, bn1, conv2, bn2, layer1, transition{1,2,3},
stage{2,3,4}.<m>.branches.<b>.<k>, stage<s>.<m>.fuse_layers.<i>.<j>..., final_layer) and the
parameter construction order (hence seeded random init) equal the reference's.  The torch.nn
layers only own parameters; ``describe`` emits the engine graph.
"""
import os

import torch.nn as nn
import yaml

from ...module import HipModule
from ..simplebaseline.pose_resnet import ResidualUnit, make_stage, describe_stage

BN_MOMENTUM = 0.1
_KIND = {"BASIC": "basic", "BOTTLENECK": "bottleneck"}          # reference blocks_dict :268-271
_EXP = {"basic": 1, "bottleneck": 4}
CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config")


def hrnet_cfg(width=48, num_joints=21):
    """The shipped config with NUM_CHANNELS = width * (1, 2, 4, 8) (the reference ships W48)."""
    with open(os.path.join(CFG_DIR, "cfg.yaml")) as f:
        cfg = yaml.safe_load(f)
    for s, n in (("STAGE2", 2), ("STAGE3", 3), ("STAGE4", 4)):
        cfg["MODEL"]["EXTRA"][s]["NUM_CHANNELS"] = [width * (2 ** i) for i in range(n)]
    cfg["MODEL"]["NUM_JOINTS"] = num_joints
    return cfg


def _conv_bn(cin, cout, k, stride, relu, momentum=None):
    bn = nn.BatchNorm2d(cout) if momentum is None else nn.BatchNorm2d(cout, momentum=momentum)
    layers = [nn.Conv2d(cin, cout, k, stride, k // 2, bias=False), bn]
    if relu:
        layers.append(nn.ReLU(True))
    return nn.Sequential(*layers)


class HighResolutionModule(nn.Module):
    """Parallel branches of residual units + the exchange (fuse) unit; reference :101-265."""

    def __init__(self, num_branches, kind, num_blocks, num_inchannels, num_channels, fuse_method,
                 multi_scale_output=True):
        super().__init__()
        for name, lst in (("NUM_BLOCKS", num_blocks), ("NUM_CHANNELS", num_channels), ("NUM_INCHANNELS", num_inchannels)):
            if num_branches != len(lst):
                raise ValueError("NUM_BRANCHES({}) <> {}({})".format(num_branches, name, len(lst)))
        self.num_inchannels = num_inchannels
        self.num_branches = num_branches
        self.multi_scale_output = multi_scale_output
        self.fuse_method = fuse_method
        branches = []
        for i in range(num_branches):
            seq, cout = make_stage(kind, num_inchannels[i], num_channels[i], num_blocks[i], 1)
            self.num_inchannels[i] = cout
            branches.append(seq)
        self.branches = nn.ModuleList(branches)
        self.fuse_layers = self._make_fuse()
        self.relu = nn.ReLU(True)

    def _make_fuse(self):
        if self.num_branches == 1:
            return None
        c = self.num_inchannels
        rows = []
        for i in range(self.num_branches if self.multi_scale_output else 1):
            row = []
            for j in range(self.num_branches):
                if j > i:        # 1x1 conv + BN + nearest upsample x2^(j-i)   (reference :196-208)
                    row.append(nn.Sequential(nn.Conv2d(c[j], c[i], 1, 1, 0, bias=False), nn.BatchNorm2d(c[i]),
                                             nn.Upsample(scale_factor=2 ** (j - i), mode="nearest")))
                elif j == i:
                    row.append(None)
                else:            # (i-j) stride-2 3x3 convs, ReLU on all but the last (reference :211-240)
                    chain = [_conv_bn(c[j], c[i] if k == i - j - 1 else c[j], 3, 2, relu=(k != i - j - 1))
                             for k in range(i - j)]
                    row.append(nn.Sequential(*chain))
            rows.append(nn.ModuleList(row))
        return nn.ModuleList(rows)

    def get_num_inchannels(self):
        return self.num_inchannels

    def describe(self, gb, prefix, xs):
        if self.num_branches == 1:
            return [describe_stage(gb, self.branches[0], f"{prefix}.branches.0", xs[0])]
        # The branches -- and the exchange convolutions that read branch j's output -- are independent chains: each runs
        # on its own stream lane between fork and join; the cross-resolution sums follow on the main lane.
        gb.fork()
        xs = list(xs)
        pre = {}                                           # (i, j) -> term of output i that comes from branch j
        for j in range(self.num_branches):
            gb.lane = j
            xs[j] = describe_stage(gb, self.branches[j], f"{prefix}.branches.{j}", xs[j])
            for i in range(len(self.fuse_layers)):
                q = f"{prefix}.fuse_layers.{i}.{j}"
                if j == i:
                    pre[(i, j)] = xs[j]
                elif j > i:
                    pre[(i, j)] = (gb.conv(xs[j], q + ".0", 1, 1, 0), q + ".1", j - i)
                else:
                    t = xs[j]
                    for k in range(i - j):
                        y = gb.conv(t, f"{q}.{k}.0", 3, 2, 1)
                        if k != i - j - 1:
                            t = gb.fuse([(y, f"{q}.{k}.1")])
                        else:
                            pre[(i, j)] = (y, f"{q}.{k}.1")
        gb.join()
        # the cross-resolution sums (pose_hrnet.py:247-265) are independent of each other: one lane each, so the plan runs
        # them -- and the BatchNorm / ReLU backward of all their terms -- as multi-problem launches
        gb.fork()
        outs = []
        for i in range(len(self.fuse_layers)):
            gb.lane = i
            outs.append(gb.fuse([pre[(i, j)] for j in range(self.num_branches)]))
        gb.join()
        return outs


class PoseHighResolutionNet(HipModule):

    def __init__(self, cfg, **kwargs):
        super().__init__()
        extra = cfg["MODEL"]["EXTRA"]
        self.conv1 = nn.Conv2d(3, 64, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.layer1, c = make_stage("bottleneck", 64, 64, 4)
        pre = [c]
        for s in (2, 3, 4):
            sc = extra[f"STAGE{s}"]
            kind = _KIND[sc["BLOCK"]]
            chans = [n * _EXP[kind] for n in sc["NUM_CHANNELS"]]
            setattr(self, f"stage{s}_cfg", sc)
            setattr(self, f"transition{s - 1}", self._make_transition(pre, chans))
            stage, pre = self._make_stage(sc, chans, multi_scale_output=(s != 4))
            setattr(self, f"stage{s}", stage)
        fk = extra["FINAL_CONV_KERNEL"]
        self.final_layer = nn.Conv2d(pre[0], cfg["MODEL"]["NUM_JOINTS"], fk, 1, 1 if fk == 3 else 0)
        self.pretrained_layers = extra["PRETRAINED_LAYERS"]

    def init_weights(self, pretrained=""):
        """pose_hrnet.py:462-492 (the reference's factory has the call commented out): normal(0, 0.001)
        conv / deconv weights, zero biases, unit BN; then, if `pretrained` is a file, load the entries whose first key
        component is listed in EXTRA.PRETRAINED_LAYERS ('*' = all) with strict=False; a non-empty path that does not
        exist raises ValueError like the reference."""
        import os
        import torch
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if os.path.isfile(pretrained):
            sd = torch.load(pretrained, map_location="cpu")
            keep = {k: v for k, v in sd.items()
                    if k.split(".")[0] in self.pretrained_layers or self.pretrained_layers[0] == "*"}
            self.load_state_dict(keep, strict=False)
        elif pretrained:
            raise ValueError("{} is not exist!".format(pretrained))

    @staticmethod
    def _make_transition(pre, cur):
        layers = []
        for i in range(len(cur)):
            if i < len(pre):
                layers.append(_conv_bn(pre[i], cur[i], 3, 1, relu=True) if cur[i] != pre[i] else None)
            else:        # new lower-resolution branch: stride-2 convs from the LAST previous branch (:358-370)
                chain = []
                for j in range(i + 1 - len(pre)):
                    cout = cur[i] if j == i - len(pre) else pre[-1]
                    chain.append(_conv_bn(pre[-1], cout, 3, 2, relu=True))
                layers.append(nn.Sequential(*chain))
        return nn.ModuleList(layers)

    @staticmethod
    def _make_stage(sc, num_inchannels, multi_scale_output=True):
        mods = []
        for m in range(sc["NUM_MODULES"]):
            mso = multi_scale_output or m != sc["NUM_MODULES"] - 1     # only the last module may drop rows
            mods.append(HighResolutionModule(sc["NUM_BRANCHES"], _KIND[sc["BLOCK"]], sc["NUM_BLOCKS"], num_inchannels,
                                             sc["NUM_CHANNELS"], sc["FUSE_METHOD"], mso))
            num_inchannels = mods[-1].get_num_inchannels()
        return nn.Sequential(*mods), num_inchannels

    def describe(self, gb):
        x = gb.input()
        x = gb.fuse([(gb.conv(x, "conv1", 3, 2, 1), "bn1")])
        x = gb.fuse([(gb.conv(x, "conv2", 3, 2, 1), "bn2")])
        x = describe_stage(gb, self.layer1, "layer1", x)
        ys = [x]
        for s in (2, 3, 4):
            trans = getattr(self, f"transition{s - 1}")
            xs = []
            for i in range(getattr(self, f"stage{s}_cfg")["NUM_BRANCHES"]):
                t = trans[i]
                if t is None:
                    xs.append(ys[i])
                elif i < len(ys):
                    xs.append(gb.fuse([(gb.conv(ys[i], f"transition{s - 1}.{i}.0", 3, 1, 1), f"transition{s - 1}.{i}.1")]))
                else:
                    a = ys[-1]
                    for j in range(len(t)):
                        q = f"transition{s - 1}.{i}.{j}"
                        a = gb.fuse([(gb.conv(a, q + ".0", 3, 2, 1), q + ".1")])
                    xs.append(a)
            for m, mod in enumerate(getattr(self, f"stage{s}")):
                xs = mod.describe(gb, f"stage{s}.{m}", xs)
            ys = xs
        fk = self.final_layer.kernel_size[0]
        gb.output(gb.conv(ys[0], "final_layer", fk, 1, 1 if fk == 3 else 0, bias="final_layer.bias"))


def get_hrnet(cfg, is_train, **kwargs):
    """Factory with the reference's signature (pose_hrnet.py:495-501)."""
    return PoseHighResolutionNet(cfg, **kwargs)
