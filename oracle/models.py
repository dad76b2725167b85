"""Oracle (CPU, plain PyTorch fp32): functional SimpleBaseline / HRNet forward from a
reference-layout ``state_dict`` + a 3-line Adam, for whole-model and trajectory parity.

TEST INFRASTRUCTURE - see oracle/__init__.py.  Nothing here is a torch.nn.Module: the
network is evaluated directly from the tensors of a state_dict whose keys follow the
reference (checkpoint row), so that the SAME dict can be fed to the
HIP engine and to this oracle.
"""
import math

import torch
import torch.nn.functional as F

BN_MOMENTUM = 0.1     # src/modeling/simplebaseline/pose_resnet.py:19, pose_hrnet.py:18
BN_EPS = 1e-5         # torch.nn.BatchNorm2d default

# src/modeling/simplebaseline/pose_resnet.py:301-305
RESNET_SPEC = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]),
               50: ("bottleneck", [3, 4, 6, 3]), 101: ("bottleneck", [3, 4, 23, 3]),
               152: ("bottleneck", [3, 8, 36, 3])}


def _bn(sd, p, x, training):
    """BatchNorm2d(momentum=0.1): batch stats + running update in training, else running stats."""
    y = F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                     sd[p + ".bias"], training, BN_MOMENTUM, BN_EPS)
    if training and (p + ".num_batches_tracked") in sd:
        sd[p + ".num_batches_tracked"] += 1
    return y


def _conv(sd, p, x, stride=1, pad=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride, pad)


def _basic(sd, p, x, stride, training):
    # pose_resnet.py:29-58 / pose_hrnet.py:28-57
    out = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, stride, 1), training))
    out = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, 1, 1), training)
    res = x
    if (p + ".downsample.0.weight") in sd:
        res = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride, 0), training)
    return F.relu(out + res)


def _bottleneck(sd, p, x, stride, training, caffe=False):
    # pose_resnet.py:61-99 (stride on the 3x3) and :102-141 (caffe: stride on the first 1x1)
    s1, s2 = (stride, 1) if caffe else (1, stride)
    out = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, s1, 0), training))
    out = F.relu(_bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, s2, 1), training))
    out = _bn(sd, p + ".bn3", _conv(sd, p + ".conv3", out, 1, 0), training)
    res = x
    if (p + ".downsample.0.weight") in sd:
        res = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride, 0), training)
    return F.relu(out + res)


def pose_resnet_forward(sd, x, num_layers=50, style="pytorch", training=True, spec=None):
    """PoseResNet.forward (pose_resnet.py:234-248) evaluated from ``sd``.  BN running
    statistics inside ``sd`` are updated in place when ``training``.  ``spec`` overrides the
    (kind, units-per-stage) lookup for truncated test networks."""
    kind, blocks = spec or RESNET_SPEC[num_layers]
    x = F.relu(_bn(sd, "bn1", _conv(sd, "conv1", x, 2, 3), training))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, n in enumerate(blocks):
        for b in range(n):
            stride = 2 if (li > 0 and b == 0) else 1
            p = f"layer{li + 1}.{b}"
            if kind == "basic":
                x = _basic(sd, p, x, stride, training)
            else:
                x = _bottleneck(sd, p, x, stride, training, caffe=(style == "caffe"))
    i = 0
    while f"deconv_layers.{3 * i}.weight" in sd:           # pose_resnet.py:207-232
        w = sd[f"deconv_layers.{3 * i}.weight"]
        k = w.shape[-1]
        pad, opad = {4: (1, 0), 3: (1, 1), 2: (0, 0)}[k]   # pose_resnet.py:194-205
        x = F.conv_transpose2d(x, w, sd.get(f"deconv_layers.{3 * i}.bias"), 2, pad, opad)
        x = F.relu(_bn(sd, f"deconv_layers.{3 * i + 1}", x, training))
        i += 1
    fk = sd["final_layer.weight"].shape[-1]
    return _conv(sd, "final_layer", x, 1, 1 if fk == 3 else 0)


# ----------------------------------------------------------------------------- HRNet
def _seq_blocks(sd, p, x, training, kind):
    b = 0
    while f"{p}.{b}.conv1.weight" in sd:
        q = f"{p}.{b}"
        x = _basic(sd, q, x, 1, training) if kind == "basic" else _bottleneck(sd, q, x, 1, training)
        b += 1
    return x


def _hr_module(sd, p, xs, training):
    """HighResolutionModule.forward (pose_hrnet.py:247-265)."""
    nb = len(xs)
    xs = [_seq_blocks(sd, f"{p}.branches.{i}", xs[i], training, "basic") for i in range(nb)]
    if nb == 1:
        return xs
    outs = []
    i = 0
    while i < nb and (i == 0 or any(k.startswith(f"{p}.fuse_layers.{i}.") for k in sd)):
        y = None
        for j in range(nb):
            if j == i:
                t = xs[j]
            elif j > i:                      # 1x1 conv + BN + nearest upsample (:196-208)
                q = f"{p}.fuse_layers.{i}.{j}"
                t = _bn(sd, q + ".1", _conv(sd, q + ".0", xs[j]), training)
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
            else:                            # chain of 3x3 s2 conv + BN (+ReLU except last) (:211-240)
                t = xs[j]
                for k in range(i - j):
                    q = f"{p}.fuse_layers.{i}.{j}.{k}"
                    t = _bn(sd, q + ".1", _conv(sd, q + ".0", t, 2, 1), training)
                    if k != i - j - 1:
                        t = F.relu(t)
            y = t if y is None else y + t
        outs.append(F.relu(y))
        i += 1
    return outs


def _transition(sd, p, ys, n_cur, training):
    """pose_hrnet.py:333-372 (build) and :434-455 (use): new branches come from ys[-1]."""
    n_pre = len(ys)
    xs = []
    for i in range(n_cur):
        if i < n_pre:
            if f"{p}.{i}.0.weight" in sd:
                xs.append(F.relu(_bn(sd, f"{p}.{i}.1", _conv(sd, f"{p}.{i}.0", ys[i], 1, 1), training)))
            else:
                xs.append(ys[i])
        else:
            t = ys[-1]
            for j in range(i + 1 - n_pre):
                q = f"{p}.{i}.{j}"
                t = F.relu(_bn(sd, q + ".1", _conv(sd, q + ".0", t, 2, 1), training))
            xs.append(t)
    return xs


def hrnet_forward(sd, x, training=True, stage_branches=(2, 3, 4)):
    """PoseHighResolutionNet.forward (pose_hrnet.py:425-460) evaluated from ``sd``."""
    x = F.relu(_bn(sd, "bn1", _conv(sd, "conv1", x, 2, 1), training))
    x = F.relu(_bn(sd, "bn2", _conv(sd, "conv2", x, 2, 1), training))
    x = _seq_blocks(sd, "layer1", x, training, "bottleneck")
    ys = [x]
    for s, nb in zip((2, 3, 4), stage_branches):
        xs = _transition(sd, f"transition{s - 1}", ys, nb, training)
        m = 0
        while any(k.startswith(f"stage{s}.{m}.") for k in sd):
            xs = _hr_module(sd, f"stage{s}.{m}", xs, training)
            m += 1
        ys = xs
    fk = sd["final_layer.weight"].shape[-1]
    return _conv(sd, "final_layer", ys[0], 1, 1 if fk == 3 else 0)


# ----------------------------------------------------------------------------- training
def clone_state(sd):
    return {k: v.detach().clone() for k, v in sd.items()}


def is_param(key):
    return key.endswith(".weight") or key.endswith(".bias")


class AdamState:
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0) restated
    (src/tools/train.py:45-48)."""

    def __init__(self, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, b1, b2, eps, 0
        self.m, self.v = {}, {}

    def step(self, sd, grads):
        self.t += 1
        c1 = 1 - self.b1 ** self.t
        c2 = 1 - self.b2 ** self.t
        for k, g in grads.items():
            m = self.m.setdefault(k, torch.zeros_like(g))
            v = self.v.setdefault(k, torch.zeros_like(g))
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(c2)).add_(self.eps)
            sd[k].addcdiv_(m, denom, value=-self.lr / c1)


def loss_and_grads(sd, forward, x, target):
    """One forward + JointsMSELoss (src/utils/loss.py:306-325) + backward through the
    functional model.  Returns (loss float, pred, {key: grad})."""
    keys = [k for k in sd if is_param(k)]
    for k in keys:
        sd[k] = sd[k].detach().requires_grad_(True)
    pred = forward(sd, x)
    loss = 0.5 * ((pred - target) ** 2).mean()
    gs = torch.autograd.grad(loss, [sd[k] for k in keys])
    for k in keys:
        sd[k] = sd[k].detach()
    return float(loss), pred.detach(), dict(zip(keys, gs))
