"""Table launches of the weight gradient (lh_wgrad_table_build / lh_wgrad_table_run; Plan._table_wgrads): the deferred weight
gradients of a group of layers as ONE grid with a split count per layer.  Reference: loss.backward(), src/utils/method.py:182,
through pose_resnet.py:61-99, 207-232.  Checked against plain PyTorch fp32 autograd on the same 16-bit operands and against the
per-layer launches (the two differ only in the fp32 summation order of the pixel splits)."""
import os

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from test_gpu_ops import TOL, rel_err, _run_plan  # noqa: E402


def _chain():
    from lighthand_amd.module import HipModule

    class ChainNet(HipModule):
        """3x3, 1x1, strided, transposed and channel-padded convolutions in a chain: tile classes 256x256 (two members), 128x128
        (two: one of them the transposed convolution, whose gradient gathers dy), and singles that stay on the per-layer path."""

        def __init__(self):
            super().__init__()
            self.c0 = nn.Conv2d(64, 128, 3, 1, 1, bias=False)
            self.c1 = nn.Conv2d(128, 256, 1, 1, 0, bias=False)
            self.c2 = nn.Conv2d(256, 256, 3, 1, 1, bias=False)
            self.c3 = nn.Conv2d(256, 256, 1, 2, 0, bias=False)
            self.d4 = nn.ConvTranspose2d(256, 128, 4, 2, 1, 0, bias=False)
            self.c5 = nn.Conv2d(128, 256, 3, 1, 1, bias=False)
            self.c6 = nn.Conv2d(256, 21, 1, 1, 0, bias=True)

        def describe(self, gb):
            x = gb.input_act(64)
            y = gb.conv(x, "c0", 3, 1, 1)
            y = gb.conv(y, "c1", 1, 1, 0)
            y = gb.conv(y, "c2", 3, 1, 1)
            y = gb.conv(y, "c3", 1, 2, 0)
            y = gb.deconv(y, "d4", 4)
            y = gb.conv(y, "c5", 3, 1, 1)
            gb.output(gb.conv(y, "c6", 1, 1, 0, bias="c6.bias"))

        def torch_forward(self, x):
            return self.c6(self.c5(self.d4(self.c3(self.c2(self.c1(self.c0(x)))))))

    return ChainNet


def _grads(proto, x, dy, precision, env, monkeypatch):
    import copy
    from lighthand_amd.engine import Plan
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    m = copy.deepcopy(proto)
    out, dx, grads = _run_plan(m, x, lambda o: dy, precision)
    plan = next(iter(m._lh_plans.values()))
    tables = [(info.n_problems, info.n_items, info.n_fold_items, info.nsplit_max, (info.bo, info.bi, info.kps, info.depth), names)
              for _, info, names in plan.wgrad_tables]
    for k in env:
        monkeypatch.delenv(k)
    del Plan
    return out, dx, grads, tables


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_table_launch_matches_torch_and_the_per_layer_launches(precision, monkeypatch):
    ChainNet = _chain()
    torch.manual_seed(11)
    proto = ChainNet()
    qt = {"bf16": torch.bfloat16, "fp16": torch.float16}[precision]
    with torch.no_grad():
        for p in proto.parameters():
            p.mul_(0.5)
            p.copy_(p.to(qt).float())
    x = torch.randn(3, 64, 12, 20).to(qt).float()
    ref = proto.torch_forward(x)
    dy = (torch.randn_like(ref) * 0.1).to(qt).float()
    monkeypatch.setenv("LH_WGRAD_GROUP", "100")                # every layer in one deferred group
    base = _grads(proto, x, dy, precision, {"LH_WGRAD_TABLE": "0"}, monkeypatch)
    assert base[3] == []
    runs = {"measured": _grads(proto, x, dy, precision, {"LH_WGRAD_TABLE_TUNE_MIN": "0"}, monkeypatch),      # (tables this small are not measured by default)
            "static": _grads(proto, x, dy, precision, {"LH_AUTOTUNE": "0"}, monkeypatch)}
    # forced configurations: split-free / short work items, 8-wave and 4-wave tiles, 32- and 64-row stages
    for tag, force in (("free", "128,128,64,3,100000"), ("short", "128,128,32,4,8"), ("small", "64,64,64,2,6")):
        runs[tag] = _grads(proto, x, dy, precision, {"LH_WGRAD_TABLE_FORCE": force, "LH_WGRAD_TABLE_BIG": "0"}, monkeypatch)
    runs["big"] = _grads(proto, x, dy, precision, {"LH_WGRAD_TABLE_FORCE": "256,256,32,3,4"}, monkeypatch)
    for tag, (out, dx, grads, tables) in runs.items():
        assert tables, tag
        tabled = {n for t in tables for n in t[5]}
        assert len(tabled) >= 4, (tag, tables)
        assert torch.equal(out, base[0]) and torch.equal(dx, base[1]), tag      # forward / data gradients do not depend on it
        for k in grads:
            e = rel_err(grads[k], base[2][k])
            assert e < 2e-5, (tag, k, e)                       # fp32 summation order of the pixel splits only
        print(tag, tables)
    # split-free 1x1 members are written by the kernel itself: a table of only such members has no fold launch
    free = runs["free"][3]
    assert all(t[3] == 1 for t in free), free
    # against autograd on the same stored operands: the activations between the layers are rounded to 16 bits on the HIP side only,
    # so compare layer by layer on the HIP side's own inputs instead -- here: the whole chain with the documented 16-bit tolerance
    ref.backward(dy)
    for k, p in proto.named_parameters():
        e = rel_err(runs["measured"][2][k], p.grad)
        assert e < 3 * TOL[precision], (k, e)


def test_table_run_is_bit_stable_and_capturable(monkeypatch):
    """The same table replayed (and replayed from a hipGraph) gives the same bits: the work-item order and every split's range
    are fixed by the table."""
    ChainNet = _chain()
    torch.manual_seed(5)
    proto = ChainNet()
    x = torch.randn(2, 64, 8, 8)
    monkeypatch.setenv("LH_WGRAD_GROUP", "100")
    monkeypatch.setenv("LH_WGRAD_TABLE_FORCE", "128,128,64,2,4")
    monkeypatch.setenv("LH_WGRAD_TABLE_BIG", "0")
    m = proto.cuda().set_precision("bf16")
    m.train()
    plan = m.plan(2, 8, 8, training=True, backward=True)
    assert plan.wgrad_tables
    plan.in_act.buf.copy_(x.permute(0, 2, 3, 1).to(plan.tdtype))
    s = torch.cuda.current_stream()
    plan.refresh_packs(s.cuda_stream)
    plan.run_forward(s.cuda_stream)
    plan.dout_nchw.normal_()
    snaps = []
    for _ in range(2):
        for g in plan.grads.values():
            g.fill_(float("nan"))
        plan.run_backward(s.cuda_stream)
        torch.cuda.synchronize()
        snaps.append({k: v.clone() for k, v in plan.grads.items()})
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(s)
    with torch.cuda.stream(side):
        for gr in plan.grads.values():
            gr.fill_(float("nan"))
        with torch.cuda.graph(g, stream=side):
            plan.run_backward(side.cuda_stream)
        g.replay()
    torch.cuda.synchronize()
    for k in snaps[0]:
        assert torch.isfinite(snaps[0][k]).all(), k
        assert torch.equal(snaps[0][k], snaps[1][k]), k
        assert torch.equal(snaps[0][k], plan.grads[k]), k
