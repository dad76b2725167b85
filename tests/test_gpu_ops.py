"""Per-kernel parity on the GPU: every HIP op vs the same op in plain PyTorch fp32 on the CPU
(golden set G4 of SURVEY.md section 8c is regenerated on the fly -- it needs no reference)."""
import ctypes as C
import os
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {"fp32": 2e-5, "bf16": 3e-2, "fp16": 4e-3}


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


_QT = {"bf16": torch.bfloat16, "fp16": torch.float16}


def quant(t, precision):
    """Round a fp32 tensor to the run precision's grid (identity for fp32): the reference side of a 16-bit case sees the
    same stored operands as the HIP side, so the tolerance covers accumulation and output rounding only."""
    return t.to(_QT[precision]).float() if precision in _QT else t


def _mods():
    from lighthand_amd.module import HipModule

    class ConvNet(HipModule):
        def __init__(self, cin, cout, k, s, p, bias=False, transposed=False):
            super().__init__()
            self.transposed, self.k, self.s, self.p, self.cin = transposed, k, s, p, cin
            if transposed:
                self.conv = nn.ConvTranspose2d(cin, cout, k, 2, {4: 1, 3: 1, 2: 0}[k], {4: 0, 3: 1, 2: 0}[k], bias=bias)
            else:
                self.conv = nn.Conv2d(cin, cout, k, s, p, bias=bias)

        def describe(self, gb):
            x = gb.input_act(self.cin)
            b = "conv.bias" if self.conv.bias is not None else None
            y = gb.deconv(x, "conv", self.k, bias=b) if self.transposed else gb.conv(x, "conv", self.k, self.s, self.p, bias=b)
            gb.output(y)

    class BnNet(HipModule):
        """conv -> BN (+ residual / second BN branch / upsampled branch) -> ReLU"""

        def __init__(self, c, mode):
            super().__init__()
            self.c, self.mode = c, mode
            self.conv = nn.Conv2d(c, c, 3, 1, 1, bias=False)
            self.bn = nn.BatchNorm2d(c, momentum=0.1)
            if mode in ("two_bn", "up"):
                self.conv2 = nn.Conv2d(c, c, 1, 2 if mode == "up" else 1, 0, bias=False)
                self.bn2 = nn.BatchNorm2d(c, momentum=0.1)
            self.out = nn.Conv2d(c, 8, 1, bias=False)

        def describe(self, gb):
            x = gb.input_act(self.c)
            y = gb.conv(x, "conv", 3, 1, 1)
            if self.mode == "plain":
                z = gb.fuse([(y, "bn")])
            elif self.mode == "residual":
                z = gb.fuse([(y, "bn"), x])
            elif self.mode == "two_bn":
                z = gb.fuse([(y, "bn"), (gb.conv(x, "conv2", 1, 1, 0), "bn2")])
            elif self.mode == "up":
                z = gb.fuse([(y, "bn"), (gb.conv(x, "conv2", 1, 2, 0), "bn2", 1)], relu=True)
            elif self.mode == "pool":
                z = gb.maxpool(gb.fuse([(y, "bn")]))
            gb.output(gb.conv(z, "out", 1, 1, 0))

        def torch_forward(self, x):
            y = self.bn(self.conv(x))
            if self.mode == "residual":
                y = y + x
            elif self.mode == "two_bn":
                y = y + self.bn2(self.conv2(x))
            elif self.mode == "up":
                y = y + F.interpolate(self.bn2(self.conv2(x)), scale_factor=2, mode="nearest")
            y = F.relu(y)
            if self.mode == "pool":
                y = F.max_pool2d(y, 3, 2, 1)
            return self.out(y)

    return ConvNet, BnNet


def _run_plan(model, x_nchw, dy_fn, precision):
    """Run fwd+bwd of a test net through the engine; returns (out, dx_nchw, {param grads})."""
    m = model.cuda().set_precision(precision)
    m.train()
    n, c, h, w = x_nchw.shape
    plan = m.plan(n, h, w, training=True, backward=True)
    plan.in_act.buf.copy_(x_nchw.permute(0, 2, 3, 1).to(plan.tdtype))
    s = torch.cuda.current_stream().cuda_stream
    plan.refresh_packs(s)
    plan.run_forward(s)
    out = plan.out_nchw.clone().cpu()
    dy = dy_fn(out)
    plan.dout_nchw.copy_(dy)
    plan.run_backward(s)
    torch.cuda.synchronize()
    dx = plan.in_act.grad.float().permute(0, 3, 1, 2).cpu()
    grads = {k: plan.grads[k].clone().cpu() for k in plan.grads}
    return out, dx, grads


CONV_CASES = [
    # cin, cout, k, s, p, n, h, w
    (64, 64, 1, 1, 0, 2, 16, 16),
    (64, 128, 3, 1, 1, 2, 16, 16),
    (128, 64, 3, 2, 1, 2, 16, 16),
    (64, 256, 1, 2, 0, 2, 16, 16),
    (32, 32, 3, 1, 1, 3, 12, 20),          # HRNet-W32 branch width, ragged pixel count
    (48, 96, 3, 2, 1, 1, 8, 8),            # HRNet-W48 widths (partial K step)
    (256, 21, 1, 1, 0, 1, 8, 8),           # padded output channels + bias path below
    (16, 16, 3, 1, 1, 1, 4, 4),
]


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bwd(case, precision):
    ConvNet, _ = _mods()
    cin, cout, k, s, p, n, h, w = case
    torch.manual_seed(1)
    m = ConvNet(cin, cout, k, s, p, bias=(cout == 21))
    x = torch.randn(n, cin, h, w)
    xq = quant(x, precision)
    ref_m = nn.Conv2d(cin, cout, k, s, p, bias=(cout == 21))
    ref_m.load_state_dict(m.conv.state_dict())
    with torch.no_grad():
        ref_m.weight.copy_(quant(ref_m.weight, precision))
    xr = xq.clone().requires_grad_(True)
    ref = ref_m(xr)
    torch.manual_seed(2)
    dy = torch.randn_like(ref)
    dyq = quant(dy, precision)
    ref.backward(dyq)
    out, dx, grads = _run_plan(m, xq, lambda o: dy, precision)
    tol = TOL[precision]
    assert rel_err(out, ref.detach()) < tol
    assert rel_err(dx, xr.grad) < tol
    assert rel_err(grads["conv.weight"], ref_m.weight.grad) < tol
    if cout == 21:
        assert rel_err(grads["conv.bias"], ref_m.bias.grad) < tol


@pytest.fixture
def forced_plans():
    """Lets a test pick the kernel configuration of every launch of the plans it builds (engine.Plan.force_cfg /
    force_wgrad: the autotuner's measurement is replaced by the test's choice)."""
    from lighthand_amd.engine import Plan
    yield Plan
    Plan.force_cfg = Plan.force_wgrad = None


CFG_CASES = [(256, 256, 3, 1, 1, 2, 16, 16), (512, 256, 1, 1, 0, 3, 12, 20), (256, 512, 3, 2, 1, 2, 16, 16), (64, 128, 3, 1, 1, 2, 20, 12)]


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("case", CFG_CASES)
def test_every_conv_kernel_configuration(case, precision, forced_plans, monkeypatch):
    """Every compiled-in configuration of the LDS-DMA convolution kernel (tile 64..256, ring depth, 64- / 128-byte
    stages: lh_igemm_candidates) that fits the launch, forced in turn on forward and data gradient: each must match
    PyTorch at the precision's tolerance (fp32 1e-3 relative is the contract; TOL is tighter), and all forms that keep ONE accumulator
    per output element agree BIT FOR BIT with one another (the K-loop order does not depend on the tile).  The K-split wave-pair forms
    (ring depth code 30..39, round 6) add two partial sums per element: they are held to PyTorch at TOL and to the other forms at the
    distance of one extra fp32 rounding of the accumulator (far inside one unit of the 16-bit output grid), not to bit equality."""
    ConvNet, _ = _mods()
    monkeypatch.setenv("LH_KSPLIT_TILES", "1")              # offer the K-split forms too (off by default: measured equal, other sum order)
    cin, cout, k, s_, p, n, h, w = case
    torch.manual_seed(5)
    x = quant(torch.randn(n, cin, h, w), precision)
    ref_m = nn.Conv2d(cin, cout, k, s_, p, bias=False)
    with torch.no_grad():
        ref_m.weight.copy_(quant(ref_m.weight, precision))
    xr = x.clone().requires_grad_(True)
    ref = ref_m(xr)
    dy = quant(torch.randn_like(ref), precision)
    ref.backward(dy)
    seen, first, ksplit_seen = [], None, False
    idx = 0
    while True:
        chosen = []

        def pick(cands, idx=idx, chosen=chosen):
            c = cands[idx % len(cands)]
            chosen.append((c, len(cands)))
            return c
        forced_plans.force_cfg = pick
        m = ConvNet(cin, cout, k, s_, p, bias=False)
        m.conv.load_state_dict(ref_m.state_dict())
        out, dx, _ = _run_plan(m, x, lambda o: dy, precision)
        assert chosen, "no launch of this plan offered candidates"
        seen.append(tuple(c for c, _ in chosen))
        assert rel_err(out, ref.detach()) < TOL[precision] and rel_err(dx, xr.grad) < TOL[precision], chosen
        ksplit = any(30 <= c[2] < 40 for c, _ in chosen)
        if first is None and not ksplit:
            first = (out, dx)
        elif ksplit:
            ksplit_seen = True
            if first is not None:       # one more fp32 rounding before the store: at most a last-place flip of a few outputs
                for a, b in ((out, first[0]), (dx, first[1])):
                    assert rel_err(a, b) < (1e-6 if precision == "fp32" else 8e-3) and float((a != b).float().mean()) < 0.05, chosen
        else:
            assert torch.equal(out, first[0]) and torch.equal(dx, first[1]), chosen
        idx += 1
        if idx >= max(nc for _, nc in chosen):
            break
    tiles = {c[0][:2] for c in seen}
    print(case, precision, len(seen), "configurations, forward tiles", sorted(tiles))
    assert len(seen) >= 3
    if precision == "bf16" and cout % 256 == 0 and out.shape[0] * out.shape[2] * out.shape[3] > 128:
        assert (256, 256) in tiles
    if precision == "bf16" and cin * k * k >= 128:
        assert ksplit_seen, "the K-split wave-pair forms were not offered"


def test_conv_tile_bn_statistics(forced_plans):
    """conv -> BN -> ReLU (+ residual) with the largest and the smallest tile forced: the tile's epilogue writes the BN
    partial sums (one slab row per pixel tile), so running statistics, BN parameter gradients and the data gradient must
    agree between the two (bf16: same stored activations, fp32 folds) and with PyTorch within the 16-bit tolerance."""
    import copy
    _, BnNet = _mods()
    for mode in ("plain", "residual"):
        torch.manual_seed(7)
        proto = BnNet(256, mode)
        x = torch.randn(2, 256, 16, 16).to(torch.bfloat16).float()
        res = {}
        for which in ("big", "small"):
            forced_plans.force_cfg = (lambda cands: max(cands, key=lambda c: (c[0] * c[1], c[3]))) if which == "big" else \
                (lambda cands: min(cands, key=lambda c: (c[0] * c[1], c[3])))
            m = copy.deepcopy(proto)
            torch.manual_seed(8)
            out, dx, grads = _run_plan(m, x, lambda o: torch.randn_like(o), "bf16")
            res[which] = (out, dx, grads, {k: v.cpu().clone() for k, v in m.state_dict().items() if "running" in k})
            plan = next(iter(m._lh_plans.values()))
            assert any("256, 256" in meta[2] for meta in plan.profile_meta if meta[2].startswith("igemm")) == (which == "big")
        a, b = res["big"], res["small"]
        assert rel_err(a[0], b[0]) < 1e-2 and rel_err(a[1], b[1]) < 2e-2
        for k in a[3]:
            assert rel_err(a[3][k], b[3][k]) < 1e-5, k
        for k in ("bn.weight", "bn.bias"):
            assert rel_err(a[2][k], b[2][k]) < 2e-2, k
        ref = copy.deepcopy(proto).train()
        y = ref.torch_forward(x)
        assert rel_err(a[0], y.detach()) < TOL["bf16"]


@pytest.mark.parametrize("case", [(256, 256, 3, 1, 1, 2, 16, 16), (64, 128, 3, 2, 1, 3, 20, 12), (512, 256, 1, 1, 0, 2, 8, 8)])
def test_every_wgrad_kernel_plan(case, forced_plans):
    """Every launch plan of the LDS-DMA weight-gradient kernel (tile, 32- / 64-row stages, ring depth, pixel splits:
    lh_wgrad_candidates) forced in turn: each matches PyTorch, all agree up to the fp32 summation order of the pixel
    splits, and a plan run twice gives the same bits."""
    ConvNet, _ = _mods()
    cin, cout, k, s_, p, n, h, w = case
    torch.manual_seed(3)
    x = torch.randn(n, cin, h, w).to(torch.bfloat16).float()
    ref_m = nn.Conv2d(cin, cout, k, s_, p, bias=False)
    with torch.no_grad():
        ref_m.weight.copy_(ref_m.weight.to(torch.bfloat16).float())
    ref = ref_m(x)
    dy = torch.randn_like(ref).to(torch.bfloat16).float()
    ref.backward(dy)
    grads, idx, ncand = [], 0, 1
    while idx < ncand:
        chosen = []

        def pick(cands, idx=idx, chosen=chosen):
            chosen.append((cands[idx % len(cands)], len(cands)))
            return cands[idx % len(cands)][:3]
        forced_plans.force_wgrad = pick
        both = []
        for _ in range(2):
            m = ConvNet(cin, cout, k, s_, p, bias=False)
            m.conv.load_state_dict(ref_m.state_dict())
            both.append(_run_plan(m, x, lambda o: dy, "bf16")[2]["conv.weight"])
        assert torch.equal(both[0], both[1]), chosen[0]
        assert rel_err(both[0], ref_m.weight.grad) < TOL["bf16"], chosen[0]
        grads.append((chosen[0][0], both[0]))
        ncand = chosen[0][1]
        idx += 1
    for cfg, g in grads[1:]:
        assert rel_err(g, grads[0][1]) < 1e-5, cfg
    tiles = {c[:2] for c, _ in grads}
    print(case, len(grads), "plans, tiles", sorted(tiles))
    assert len(grads) >= 6
    if cin % 256 == 0 and cout % 256 == 0:
        assert (256, 256) in tiles


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("case", [(64, 32, 4, 2, 8, 8), (128, 64, 4, 1, 6, 10), (32, 32, 3, 1, 4, 4), (32, 16, 2, 1, 4, 4)])
def test_deconv_fwd_bwd(case, bias, precision):
    """Transposed convolution, forward / data gradient / weight gradient -- and with ``bias`` (DECONV_WITH_BIAS=True,
    pose_resnet.py:149,227) the bias in the epilogue and its gradient from the NHWC channel-sum kernel (lh_channel_sum_nhwc)."""
    ConvNet, _ = _mods()
    cin, cout, k, n, h, w = case
    torch.manual_seed(3)
    m = ConvNet(cin, cout, k, 2, 1, bias=bias, transposed=True)
    x = torch.randn(n, cin, h, w)
    xq = quant(x, precision)
    wt = quant(m.conv.weight.detach().clone(), precision)
    wt.requires_grad_(True)
    b = None
    if bias:
        with torch.no_grad():
            m.conv.bias.uniform_(-1.0, 1.0)
        b = m.conv.bias.detach().clone().requires_grad_(True)
    xr = xq.clone().requires_grad_(True)
    pad, opad = {4: (1, 0), 3: (1, 1), 2: (0, 0)}[k]
    ref = F.conv_transpose2d(xr, wt, b, 2, pad, opad)
    torch.manual_seed(4)
    dy = torch.randn_like(ref)
    dyq = quant(dy, precision)
    ref.backward(dyq)
    out, dx, grads = _run_plan(m, xq, lambda o: dy, precision)
    tol = TOL[precision]
    assert out.shape == ref.shape
    assert rel_err(out, ref.detach()) < tol
    assert rel_err(dx, xr.grad) < tol
    assert rel_err(grads["conv.weight"], wt.grad) < tol
    if bias:
        assert rel_err(grads["conv.bias"], b.grad) < 1e-5       # fp64 sums of the same (quantised) gradient values


@pytest.mark.parametrize("mode", ["plain", "residual", "two_bn", "up", "pool"])
def test_bn_fuse_fwd_bwd_fp32(mode):
    _, BnNet = _mods()
    torch.manual_seed(5)
    m = BnNet(32, mode)
    with torch.no_grad():
        m.bn.weight.uniform_(0.5, 1.5)
        m.bn.bias.uniform_(-0.5, 0.5)
    import copy
    ref = copy.deepcopy(m)
    x = torch.randn(3, 32, 8, 12)
    xr = x.clone().requires_grad_(True)
    ref.train()
    y = ref.torch_forward(xr)
    torch.manual_seed(6)
    dy = torch.randn_like(y)
    y.backward(dy)
    out, dx, grads = _run_plan(m, x, lambda o: dy, "fp32")
    assert rel_err(out, y.detach()) < 5e-5
    assert rel_err(dx, xr.grad) < 2e-4
    rp = dict(ref.named_parameters())
    for k, g in grads.items():
        assert rel_err(g, rp[k].grad) < 3e-4, k
    # running statistics and the batch counter follow nn.BatchNorm2d(momentum=0.1)
    msd = {k: v.cpu() for k, v in m.state_dict().items()}
    for k, v in ref.state_dict().items():
        if "running" in k:
            assert rel_err(msd[k], v) < 1e-5, k
        if "num_batches" in k:
            assert int(msd[k]) == int(v) == 1


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("mode", ["plain", "residual", "two_bn", "up"])
def test_bn_backward_fold_in_apply_equals_fold_launch(mode, precision, monkeypatch):
    """Small tensors: the BN-backward apply pass folds the reduce pass's partial sums itself (bn.hip fold_coef_block, no
    coefficient launch); LH_FOLD_IN_APPLY=0 forces the separate fold launch.  Both paths sum the same partial sums in fp64,
    in two fixed orders: input gradients and d(gamma) / d(beta) agree to the last fp32 digits (pose_resnet.py:45-47,
    pose_hrnet.py:247-265 are the nodes this serves)."""
    _, BnNet = _mods()
    import copy
    torch.manual_seed(5)
    m0 = BnNet(64, mode)
    with torch.no_grad():
        m0.bn.weight.uniform_(0.5, 1.5)
        m0.bn.bias.uniform_(-0.5, 0.5)
    x = torch.randn(4, 64, 16, 24)
    torch.manual_seed(6)
    dy = None
    res = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("LH_FOLD_IN_APPLY", "0")
        m = copy.deepcopy(m0)
        out, dx, grads = _run_plan(m, x, lambda o: torch.randn(o.shape, generator=torch.Generator().manual_seed(7)), precision)
        res.append((out, dx, grads))
    (o1, d1, g1), (o2, d2, g2) = res
    assert torch.equal(o1, o2)
    tol = 1e-6 if precision == "fp32" else 1e-2          # bf16: a coefficient that moves by one fp32 ulp may move a rounding
    assert rel_err(d1, d2) < tol
    for k in g1:
        assert rel_err(g1[k], g2[k]) < (1e-6 if precision == "fp32" else 1e-2), k


def test_maxpool_ties_route_to_first():
    """All-equal windows (post-ReLU zeros): gradient goes to the first element in scan order."""
    from lighthand_amd import _lib
    lib = _lib.load()
    x = torch.zeros(1, 6, 6, 8, device="cuda")
    x[0, 2, 3, :] = 2.0
    out = torch.empty(1, 3, 3, 8, device="cuda")
    idx = torch.empty(1, 3, 3, 8, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.lh_maxpool3x3s2_fwd(x.data_ptr(), out.data_ptr(), idx.data_ptr(), 1, 6, 6, 8, _lib.LH_F32, s))
    dy = torch.ones_like(out)
    dx = torch.empty_like(x)
    _lib.check(lib.lh_maxpool3x3s2_bwd(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), 1, 6, 6, 8, _lib.LH_F32, s))
    xr = x.permute(0, 3, 1, 2).cpu().clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    yr.backward(torch.ones_like(yr))
    assert torch.equal(out.permute(0, 3, 1, 2).cpu(), yr.detach())
    assert torch.equal(dx.permute(0, 3, 1, 2).cpu(), xr.grad)


@pytest.mark.parametrize("shape", [(3, 64, 34, 30), (2, 128, 17, 23), (1, 64, 8, 8)])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_bn_relu_pool_as_one_pass_matches_the_two_launches(shape, dtype):
    """lh_bn_relu_maxpool3x3s2_fwd -- maxpool(relu(bn(x))) of the training stem as one pass over the RAW BatchNorm input
    (pose_resnet.py:153-156): pooled values AND window positions equal lh_fuse_fwd's stored activation (mul, add, max in
    fp32, one rounding) followed by lh_maxpool3x3s2_fwd, bit for bit, on ragged sizes with ties (post-ReLU zeros); the plain
    pool agrees with PyTorch's, NaNs included."""
    from lighthand_amd import _lib
    lib = _lib.load()
    n, c, h, w = shape
    td = {"bf16": torch.bfloat16, "fp16": torch.float16}[dtype]
    dt = {"bf16": _lib.LH_BF16, "fp16": _lib.LH_F16}[dtype]
    torch.manual_seed(4)
    raw = torch.randn(n, h, w, c).to(td).cuda()
    scale = (0.5 + torch.rand(c)).cuda()
    shift = (0.3 * torch.randn(c)).cuda()
    v = raw.float() * scale + shift
    act = torch.where(v > 0, v, torch.zeros_like(v)).to(td)               # what lh_fuse_fwd stores
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    s = torch.cuda.current_stream().cuda_stream
    out, idx = torch.empty(n, ho, wo, c, dtype=td, device="cuda"), torch.empty(n, ho, wo, c, dtype=torch.uint8, device="cuda")
    _lib.check(lib.lh_maxpool3x3s2_fwd(act.data_ptr(), out.data_ptr(), idx.data_ptr(), n, h, w, c, dt, s))
    out2, idx2 = torch.empty_like(out), torch.empty_like(idx)
    _lib.check(lib.lh_bn_relu_maxpool3x3s2_fwd(raw.data_ptr(), scale.data_ptr(), shift.data_ptr(), out2.data_ptr(), idx2.data_ptr(), n, h, w, c, dt, s))
    act_nan = act.clone()
    act_nan[0, 3, 5, :7] = float("nan")
    out3, idx3 = torch.empty_like(out), torch.empty_like(idx)
    _lib.check(lib.lh_maxpool3x3s2_fwd(act_nan.data_ptr(), out3.data_ptr(), idx3.data_ptr(), n, h, w, c, dt, s))
    torch.cuda.synchronize()
    assert torch.equal(out, out2) and torch.equal(idx, idx2)
    ref = F.max_pool2d(act_nan.float().permute(0, 3, 1, 2).cpu(), 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.equal(torch.nan_to_num(out3.float().cpu(), nan=-7.0), torch.nan_to_num(ref, nan=-7.0))


def test_gaussian_target_matches_oracle_and_golden(golden_dir):
    from lighthand_amd.heatmap import render_targets, generate_target
    from oracle.heatmap import generate_target as oracle_target
    import os
    g = np.load(os.path.join(golden_dir, "g1_target.npz"))
    joints = torch.from_numpy(g["joints"]).cuda()
    got = render_targets(joints).cpu().numpy()
    want = np.stack([oracle_target(j) for j in g["joints"]])
    assert np.array_equal(got, want)                       # bit-exact vs the oracle on this host
    assert np.abs(got - g["target"]).max() <= 6e-8         # golden: equal up to the host's expf ulp
    assert np.array_equal(got != 0, g["target"] != 0)      # identical support
    assert np.array_equal(generate_target(g["joints"][0]).numpy(), want[0])


def test_generate_heatmap_alt_matches_oracle_and_golden(golden_dir):
    """GenerateHeatmap (src/utils/dataset_loader.py:22-53) on the device: sigma = res / 64, 9 x 9 patch, maximum blend,
    points in heat-map coordinates, the x > 0 / inside-the-map skip rules -- against the golden captured from the
    reference (g1_target.npz["alt"] = GenerateHeatmap(64, 21)(joints / 4)) and the numpy oracle, incl. crafted edge points."""
    import os
    from lighthand_amd.heatmap import GenerateHeatmap
    from oracle.heatmap import generate_heatmap_alt
    g = np.load(os.path.join(golden_dir, "g1_target.npz"))
    gh = GenerateHeatmap(64, 21)
    pts = torch.from_numpy(g["joints"] / 4).cuda()
    got = gh.render(pts).cpu().numpy()
    assert got.shape == g["alt"].shape
    assert np.abs(got - g["alt"]).max() <= 6e-8           # golden: equal up to the host's exp ulp
    assert np.array_equal(got != 0, g["alt"] != 0)        # identical support (skip rules, clipping)
    want = np.stack([generate_heatmap_alt(p) for p in g["joints"] / 4])
    assert np.array_equal(got, want)                      # bit-exact vs the oracle on this host
    edge = np.zeros((21, 2), np.float32)
    edge[:9] = [(0.0, 5.0), (0.5, 5.0), (-3.0, 5.0), (63.9, 63.9), (64.0, 10.0), (10.0, 64.5), (1.0, 0.0), (3.2, -0.5), (62.0, 1.0)]
    edge[9:] = np.random.RandomState(0).uniform(-4, 68, size=(12, 2))
    assert np.array_equal(gh(edge), generate_heatmap_alt(edge))       # per-sample form with the reference's call signature
    assert gh(edge)[0].max() == 0 and gh(edge)[1].max() > 0          # x = 0 is skipped, x = 0.5 is drawn (pt[0] > 0, int() = 0)


def test_mse_loss_golden(golden_dir):
    import os
    from lighthand_amd.heatmap import JointsMSELoss
    g = np.load(os.path.join(golden_dir, "g2_loss.npz"))
    for tag in ("b4", "b1"):
        pred = torch.from_numpy(g[f"pred_{tag}"]).cuda().requires_grad_(True)
        tgt = torch.from_numpy(g[f"tgt_{tag}"]).cuda()
        loss = JointsMSELoss(False)(pred, tgt, None)
        loss.backward()
        assert abs(float(loss) - float(g[f"loss_{tag}"])) < 1e-6 * float(g[f"loss_{tag}"])
        assert rel_err(pred.grad.cpu(), torch.from_numpy(g[f"grad_{tag}"])) < 1e-6


def test_argmax_decode_golden_bit_exact(golden_dir):
    import os
    from lighthand_amd.heatmap import get_max_preds, max_preds_device
    g = np.load(os.path.join(golden_dir, "g3_decode.npz"))
    for hm, p, m in ((g["hm"], g["preds"], g["maxvals"]), (g["hm2"], g["preds2"], g["maxvals2"])):
        preds, maxvals = get_max_preds(hm)
        assert np.array_equal(preds, p)
        assert np.array_equal(maxvals, m, equal_nan=True)
        pd, md, idx = max_preds_device(torch.from_numpy(hm).cuda(), scale=4.0)
        assert np.array_equal(pd.cpu().numpy(), p * 4)


def test_quarter_pixel_refinement_opt_in(golden_dir):
    """SURVEY 8f rank 4, an EXTENSION (no reference oracle: TEST.POST_PROCESS exists in the reference's config but is
    never read).  Checked bit-exactly against the numpy restatement of the published SimpleBaseline rule on the G3
    heatmaps (ties, NaN, all-negative, corner peaks -> no move) and on smooth Gaussians (-> +-0.25 moves)."""
    import os
    from lighthand_amd.heatmap import get_max_preds, max_preds_device, render_targets
    from oracle.heatmap import get_max_preds as oracle_max, refine_quarter_pixel
    g = np.load(os.path.join(golden_dir, "g3_decode.npz"))
    rng = np.random.RandomState(5)
    joints = torch.from_numpy(rng.uniform(8, 248, size=(4, 21, 2)).astype(np.float32)).cuda()
    smooth = render_targets(joints).cpu().numpy() + 0.01 * rng.rand(4, 21, 64, 64).astype(np.float32)
    moved = 0
    for hm in (g["hm"], g["hm2"], smooth):
        p0, m0 = oracle_max(hm)
        want = refine_quarter_pixel(hm, p0, m0)
        got, _ = get_max_preds(hm, post_process=True)
        assert np.array_equal(got, want)
        moved += int((want != p0).any(-1).sum())
        dev, _, _ = max_preds_device(torch.from_numpy(hm).cuda(), scale=4.0, post_process=True)
        assert np.array_equal(dev.cpu().numpy(), want * 4)
        off, _ = get_max_preds(hm)                       # default stays the reference's hard arg-max
        assert np.array_equal(off, p0)
    assert moved > 50


def test_soft_argmax_opt_in():
    """Soft-arg-max decode (named in the brief, absent from the reference: extension without a reference oracle) against
    its float64 numpy restatement; on sharp single peaks it coincides with the hard arg-max."""
    from lighthand_amd.heatmap import get_max_preds, render_targets, soft_argmax_device
    from oracle.heatmap import soft_argmax
    rng = np.random.RandomState(2)
    hm = rng.randn(3, 21, 48, 64).astype(np.float32)
    for beta in (1.0, 25.0):
        got = soft_argmax_device(torch.from_numpy(hm).cuda(), beta=beta, scale=4.0).cpu().numpy()
        assert np.allclose(got, soft_argmax(hm, beta) * 4, rtol=1e-5, atol=1e-4)
    joints = torch.from_numpy(rng.uniform(40, 216, size=(2, 21, 2)).astype(np.float32)).cuda()
    peaks = render_targets(joints)
    hard, _ = get_max_preds(peaks)
    soft = soft_argmax_device(peaks, beta=200.0)
    assert float((soft - hard).abs().max()) < 0.51


def test_adam_matches_torch():
    from lighthand_amd.optim import Adam
    torch.manual_seed(7)
    p0 = torch.randn(1000 + 3)
    pr = p0.clone().requires_grad_(True)
    pg = p0.clone().cuda().requires_grad_(True)
    o_ref = torch.optim.Adam([pr], lr=1e-3)
    o_hip = Adam([pg], lr=1e-3)
    for i in range(5):
        g = torch.randn(1003) * (10.0 ** (i - 2))
        pr.grad = g.clone()
        pg.grad = g.clone().cuda()
        o_ref.step()
        o_hip.step()
    assert rel_err(pg.detach().cpu(), pr.detach()) < 1e-6


@pytest.mark.parametrize("src", [(224, 224), (256, 256), (300, 200)])
def test_uint8_input_pipeline_matches_torch_transform_chain(src):
    """uint8 HWC -> /255 -> bilinear resize -> Normalize, fused on the device, vs the same chain in
    plain PyTorch on the CPU (ToTensor, F.interpolate bilinear half-pixel, Normalize; dataset.py:128-159)."""
    import torch.nn.functional as F
    from conftest import resnet_cfg
    from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
    torch.manual_seed(0)
    m = get_pose_net(resnet_cfg(18), True).cuda()
    plan = m.plan(2, 64, 64, training=False, backward=False)
    hs, ws = src
    hs, ws = hs // 4, ws // 4                    # small: the target is the 64 x 64 test plan
    u8 = torch.randint(0, 256, (2, hs, ws, 3), dtype=torch.uint8)
    buf = plan.use_uint8_input(hs, ws)
    buf.copy_(u8)
    s = torch.cuda.current_stream().cuda_stream
    plan.fwd[plan._image_call_index](s)
    torch.cuda.synchronize()
    got = plan.img_nhwc4.float().cpu()           # [n][h+2p][wp][4]
    x = u8.permute(0, 3, 1, 2).float() / 255.0
    x = F.interpolate(x, size=(64, 64), mode="bilinear", align_corners=False, antialias=False)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    want = ((x - mean) / std).permute(0, 2, 3, 1)
    p = plan.img_pad
    assert torch.allclose(got[:, p:p + 64, p:p + 64, :3], want, atol=2e-5, rtol=1e-5)
    assert float(got[:, :p].abs().max()) == 0 and float(got[..., 3].abs().max()) == 0      # zero border and pad channel
    # the model consumes it: same heatmaps as feeding the float tensor
    plan.refresh_packs(s)
    plan.run_forward(s)
    a = plan.out_nchw.clone()
    m2_out = m.eval()(want.permute(0, 3, 1, 2).contiguous().cuda())
    assert torch.allclose(a, m2_out, atol=1e-4, rtol=1e-3)


def test_color_jitter_input_pipeline_matches_oracle():
    """The training transform chain with ColorJitter (dataset.py:134-146) fused on the device: per-image factors and
    op orders (all 24 permutations, skipped images, factors at the ends of the reference's 0.5 ranges) against the
    numpy restatement of torchvision's published tensor algorithm (oracle/color.py)."""
    import itertools
    from conftest import resnet_cfg
    from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
    from lighthand_amd.runtime import sample_color_jitter
    from oracle import color as oc
    torch.manual_seed(0)
    perms = list(itertools.permutations(range(4)))
    n, hs, ws = len(perms) + 4, 56, 40
    m = get_pose_net(resnet_cfg(18), True).cuda()
    plan = m.plan(n, 64, 64, training=False, backward=False)
    rng = np.random.RandomState(3)
    u8 = rng.randint(0, 256, size=(n, hs, ws, 3)).astype(np.uint8)
    u8[1] = (u8[1] // 4) * 4 // 4 * 4                              # an image with many equal channels (grey pixels: hue no-op)
    u8[1][..., 1] = u8[1][..., 0]
    u8[1][..., 2] = u8[1][..., 0]
    factors, order = sample_color_jitter(n, generator=torch.Generator().manual_seed(5))
    for i, p in enumerate(perms):
        order[i] = torch.tensor(p, dtype=torch.int32)
    order[len(perms)] = torch.tensor([-1, -1, -1, -1], dtype=torch.int32)                 # not jittered at all
    order[len(perms) + 1] = torch.tensor([3, -1, 1, -1], dtype=torch.int32)               # a subset of the ops
    factors[len(perms) + 2] = torch.tensor([0.5, 0.5, 0.5, -0.5])                          # range ends
    factors[len(perms) + 3] = torch.tensor([1.5, 1.5, 1.5, 0.5])
    buf = plan.use_uint8_input(hs, ws, jitter=True)
    buf.copy_(torch.from_numpy(u8))
    plan.jitter_factors.copy_(factors)
    plan.jitter_order.copy_(order)
    s = torch.cuda.current_stream().cuda_stream
    plan.fwd[plan._image_call_index](s)
    torch.cuda.synchronize()
    got = plan.img_nhwc4.float().cpu().numpy()
    p = plan.img_pad
    worst = 0.0
    for i in range(n):
        want = oc.input_pipeline(u8[i], 64, 64, factors[i].numpy(), [int(v) for v in order[i]])
        g = np.transpose(got[i, p:p + 64, p:p + 64, :3], (2, 0, 1))
        d = np.abs(g - want)
        # a pixel on a hue-sector / clamp boundary may take the other branch after fp32 reassociation: allow a handful
        assert (d > 1e-4).mean() < 1e-3, (i, order[i], float(d.max()))
        worst = max(worst, float(np.median(d)))
    assert worst < 1e-6
    assert float(np.abs(got[:, :p]).max()) == 0 and float(np.abs(got[..., 3]).max()) == 0


def test_device_metrics_match_host_metrics(golden_dir):
    import json, os
    from lighthand_amd import metrics as M
    g = json.load(open(os.path.join(golden_dir, "g7_metrics.json")))
    pred, gt = torch.tensor(g["val_pred"]), torch.tensor(g["val_gt"])
    pck, esum, ecnt = M.device_pck_epe(pred.cuda(), gt.cuda(), T=0.2)
    assert abs(float(pck) - g["pck02"]) < 1e-6
    assert abs(float(esum) - g["epe_sum"]) < 1e-4 * g["epe_sum"] and float(ecnt) == g["epe_cnt"]
    rng = np.random.RandomState(2)
    p2 = torch.from_numpy(rng.uniform(0, 256, (37, 21, 2)).astype(np.float32))
    g2 = torch.from_numpy(rng.uniform(20, 236, (37, 21, 3)).astype(np.float32))
    pck, esum, ecnt = M.device_pck_epe(p2.cuda(), g2.cuda(), T=0.5)
    assert abs(float(pck) - M.PCK_2d_loss(p2, g2, T=0.5)) < 1e-6
    (s, c), _ = M.EPE_train(p2, g2)
    assert abs(float(esum) - s) < 1e-4 * s and float(ecnt) == c


def test_device_pck_curve_equals_pred_eval():
    """pred_eval's per-category PCK curve / AUC / EPE (src/utils/argparser.py:326-388) counted on the
    device, accumulated over batches; integer counts, so the curve is EXACTLY the host one and AUC / EPE agree to 1e-12."""
    from lighthand_amd.metrics import auc_from_counts, device_pck_curve, pred_eval
    rng = np.random.RandomState(8)
    n, j = 37, 21
    gt = np.concatenate([rng.uniform(20, 236, (n, j, 2)), (rng.rand(n, j, 1) < 0.7).astype(np.float64)], -1).astype(np.float32)
    pred = (gt[..., :2] + rng.randn(n, j, 2) * 9).astype(np.float32)
    bb = rng.uniform(60, 200, n).astype(np.float32)
    meta = {"cat": {"bb": bb.astype(np.float64).tolist(), "pred": pred.astype(np.float64).tolist(), "gt": gt.astype(np.float64).tolist()}}
    for T_list, method in (([0.1, 0.3], "pckb"), ([0, 30], "mm"), ([0, 50], "mm")):
        want = pred_eval(meta, T_list, method)["cat"]
        acc = None
        for lo, hi in ((0, 16), (16, 37)):              # two batches, accumulated on the device
            acc = device_pck_curve(torch.from_numpy(pred[lo:hi]).cuda(), torch.from_numpy(gt[lo:hi]).cuda(),
                                   torch.from_numpy(bb[lo:hi]).cuda(), T_list, method, out=acc)
        counts, nvis, diff_sum, n_all = (t.cpu().numpy() for t in acc)
        got = auc_from_counts(counts, nvis[0], diff_sum[0], n_all[0], T_list, method)
        assert np.array_equal(got[2], want[2])
        assert abs(got[0] - want[0]) < 1e-12 and abs(got[1] - want[1]) < 1e-9 * max(1.0, abs(want[1]))


PW_CASES = [
    # cin, cout, stride, n, h, w -- the 1x1 layers of the bottleneck stages (pose_resnet.py:66-72, 180-184), the head, HRNet widths
    (64, 256, 1, 2, 16, 16), (256, 64, 1, 2, 16, 16), (64, 64, 1, 1, 5, 7), (256, 128, 1, 3, 12, 20),
    (128, 512, 1, 2, 8, 8), (512, 128, 1, 2, 8, 8), (256, 512, 2, 2, 16, 16), (256, 21, 1, 1, 8, 8), (96, 48, 1, 1, 9, 9),
    (256, 1024, 1, 1, 8, 8),
]


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("case", PW_CASES)
def test_pointwise_kernel_configurations(case, precision, forced_plans):
    """Every configuration of the persistent pointwise kernel (igemm_pw_kernel.h: resident weight panel 64..256 channels,
    16..64 pixels per wave step) that fits a 1x1 launch, forced on forward and data gradient: bit-equal with the tiled
    LDS-DMA kernel (same K order, same epilogue arithmetic) and within tolerance of PyTorch."""
    ConvNet, _ = _mods()
    cin, cout, s_, n, h, w = case
    torch.manual_seed(11)
    bias = cout == 21
    x = quant(torch.randn(n, cin, h, w), precision)
    ref_m = nn.Conv2d(cin, cout, 1, s_, 0, bias=bias)
    with torch.no_grad():
        ref_m.weight.copy_(quant(ref_m.weight, precision))
    xr = x.clone().requires_grad_(True)
    ref = ref_m(xr)
    dy = quant(torch.randn_like(ref), precision)
    ref.backward(dy)

    def run(pick):
        forced_plans.force_cfg = pick
        m = ConvNet(cin, cout, 1, s_, 0, bias=bias)
        m.conv.load_state_dict(ref_m.state_dict())
        return _run_plan(m, x, lambda o: dy, precision)

    base = run(lambda cands: next(c for c in cands if c[2] not in (1, 100)))        # a tiled configuration
    assert rel_err(base[0], ref.detach()) < TOL[precision] and rel_err(base[1], xr.grad) < TOL[precision]
    npw, idx = 1, 0
    seen = set()
    while idx < npw:
        chosen = []

        def pick(cands, idx=idx, chosen=chosen):
            pw = [c for c in cands if c[2] == 1]
            chosen.append((pw[idx % len(pw)] if pw else None, len(pw)))
            return pw[idx % len(pw)] if pw else cands[0]
        out, dx, _ = run(pick)
        assert torch.equal(out, base[0]) and torch.equal(dx, base[1]), (case, chosen)
        seen |= {c for c, _ in chosen if c}
        npw = max(nc for _, nc in chosen)
        idx += 1
    print(case, precision, "pointwise configurations:", sorted(seen))
    assert seen, "no pointwise configuration was offered"


@pytest.mark.parametrize("case", [(64, 256, 2, 16, 16), (256, 64, 3, 12, 20), (128, 512, 1, 9, 9), (512, 128, 2, 8, 8)])
def test_pointwise_kernel_bn_statistics(case, forced_plans):
    """1x1 conv -> BN -> ReLU -> 1x1 with the pointwise kernel forced: it writes ONE statistics row per workgroup (sums
    kept in registers over all its tiles).  Outputs equal the tiled kernel's bit for bit; running statistics and BN
    gradients agree to fp32 summation order; everything within the 16-bit tolerance of PyTorch."""
    import copy
    from lighthand_amd.module import HipModule
    cin, cout, n, h, w = case

    class Net(HipModule):
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv2d(cin, cout, 1, bias=False)
            self.bn = nn.BatchNorm2d(cout, momentum=0.1)
            self.out = nn.Conv2d(cout, 8, 1, bias=False)

        def describe(self, gb):
            x = gb.input_act(cin)
            gb.output(gb.conv(gb.fuse([(gb.conv(x, "conv", 1, 1, 0), "bn")]), "out", 1, 1, 0))

        def torch_forward(self, x):
            return self.out(F.relu(self.bn(self.conv(x))))

    torch.manual_seed(21)
    proto = Net()
    with torch.no_grad():
        for p_ in proto.parameters():
            p_.copy_(p_.to(torch.bfloat16).float())
    x = torch.randn(n, cin, h, w).to(torch.bfloat16).float()
    res = {}
    for which in ("tiled", "pw"):
        if which == "tiled":
            forced_plans.force_cfg = lambda cands: next(c for c in cands if c[2] not in (1, 100))
        else:
            forced_plans.force_cfg = lambda cands: max([c for c in cands if c[2] == 1] or cands[:1], key=lambda c: (c[2] == 1, c[0], c[1]))
        m = copy.deepcopy(proto)
        torch.manual_seed(22)
        out, dx, grads = _run_plan(m, x, lambda o: torch.randn_like(o), "bf16")
        res[which] = (out, dx, grads, {k: v.cpu().clone() for k, v in m.state_dict().items() if "running" in k})
        plan = next(iter(m._lh_plans.values()))
        assert any(meta[2].startswith("igemm_pw_kernel") and meta[2].endswith("true>") for meta in plan.profile_meta) == (which == "pw")
    a, b = res["tiled"], res["pw"]
    for k in a[3]:
        assert rel_err(a[3][k], b[3][k]) < 1e-5, k
    assert rel_err(a[0], b[0]) < 1e-2 and rel_err(a[1], b[1]) < 2e-2
    for k in ("bn.weight", "bn.bias"):
        assert rel_err(a[2][k], b[2][k]) < 2e-2, k
    ref = copy.deepcopy(proto).train()
    assert rel_err(b[0], ref.torch_forward(x).detach()) < TOL["bf16"]


D3_CASES = [
    # cin, cout, n, h, w -- 3x3 / stride 1 / padding 1 with 32 or 64 input channels (pose_resnet.py:66-72 stage 1, pose_hrnet.py:139-185)
    (64, 64, 2, 16, 16), (64, 64, 3, 12, 20), (64, 128, 1, 9, 9), (32, 32, 2, 16, 32), (32, 64, 1, 7, 5), (64, 32, 2, 8, 16), (64, 256, 1, 8, 8),
]


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("case", D3_CASES)
def test_direct3x3_kernel_configurations(case, precision, forced_plans):
    """The direct 3x3 kernel (conv3x3_direct_kernel.h: LDS-resident weights, one input patch per 8 x 16 tile, taps =
    offsets into the patch) forced on forward and data gradient: bit-equal with the tiled LDS-DMA kernel (same tap-major
    K order, same epilogue arithmetic) incl. ragged maps (partial tiles, zero padding at every border), several output
    channel blocks, and within tolerance of PyTorch."""
    ConvNet, _ = _mods()
    cin, cout, n, h, w = case
    torch.manual_seed(13)
    x = quant(torch.randn(n, cin, h, w), precision)
    ref_m = nn.Conv2d(cin, cout, 3, 1, 1, bias=False)
    with torch.no_grad():
        ref_m.weight.copy_(quant(ref_m.weight, precision))
    xr = x.clone().requires_grad_(True)
    ref = ref_m(xr)
    dy = quant(torch.randn_like(ref), precision)
    ref.backward(dy)

    def run(pick):
        forced_plans.force_cfg = pick
        m = ConvNet(cin, cout, 3, 1, 1, bias=False)
        m.conv.load_state_dict(ref_m.state_dict())
        return _run_plan(m, x, lambda o: dy, precision)

    base = run(lambda cands: next(c for c in cands if c[2] not in (1, 100)))
    assert rel_err(base[0], ref.detach()) < TOL[precision] and rel_err(base[1], xr.grad) < TOL[precision]
    used = []

    def pick(cands):
        d3 = [c for c in cands if c[2] == 100]
        used.append(bool(d3))
        return d3[0] if d3 else cands[0]
    out, dx, _ = run(pick)
    assert torch.equal(out, base[0]) and torch.equal(dx, base[1]), case
    # forward: cin in {32, 64}; data gradient: its K is cout
    assert used[0] and (len(used) < 2 or used[1] == (cout in (32, 64)))


@pytest.mark.parametrize("case", [(64, 64, 2, 16, 16), (32, 32, 3, 12, 20), (64, 128, 1, 9, 9)])
def test_direct3x3_kernel_bn_statistics(case, forced_plans):
    """3x3 conv -> BN -> ReLU -> 1x1 with the direct kernel forced: one statistics row per workgroup.  Outputs equal the
    tiled kernel's within the 16-bit tolerance, running statistics and BN gradients agree to fp32 summation order."""
    import copy
    from lighthand_amd.module import HipModule
    cin, cout, n, h, w = case

    class Net(HipModule):
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=False)
            self.bn = nn.BatchNorm2d(cout, momentum=0.1)
            self.out = nn.Conv2d(cout, 8, 1, bias=False)

        def describe(self, gb):
            x = gb.input_act(cin)
            gb.output(gb.conv(gb.fuse([(gb.conv(x, "conv", 3, 1, 1), "bn")]), "out", 1, 1, 0))

        def torch_forward(self, x):
            return self.out(F.relu(self.bn(self.conv(x))))

    torch.manual_seed(23)
    proto = Net()
    with torch.no_grad():
        for p_ in proto.parameters():
            p_.copy_(p_.to(torch.bfloat16).float())
    x = torch.randn(n, cin, h, w).to(torch.bfloat16).float()
    res = {}
    for which in ("tiled", "direct"):
        if which == "tiled":
            forced_plans.force_cfg = lambda cands: next(c for c in cands if c[2] not in (1, 100))
        else:
            forced_plans.force_cfg = lambda cands: next((c for c in cands if c[2] == 100), cands[0])
        m = copy.deepcopy(proto)
        torch.manual_seed(24)
        out, dx, grads = _run_plan(m, x, lambda o: torch.randn_like(o), "bf16")
        res[which] = (out, dx, grads, {k: v.cpu().clone() for k, v in m.state_dict().items() if "running" in k})
        plan = next(iter(m._lh_plans.values()))
        assert any(meta[2].startswith("conv3x3_direct_kernel") and meta[2].endswith("true>") for meta in plan.profile_meta) == (which == "direct")
    a, b = res["tiled"], res["direct"]
    for k in a[3]:
        assert rel_err(a[3][k], b[3][k]) < 1e-5, k
    assert rel_err(a[0], b[0]) < 1e-2 and rel_err(a[1], b[1]) < 2e-2
    for k in ("bn.weight", "bn.bias"):
        assert rel_err(a[2][k], b[2][k]) < 2e-2, k
    ref = copy.deepcopy(proto).train()
    assert rel_err(b[0], ref.torch_forward(x).detach()) < TOL["bf16"]


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_short_k_run_does_not_read_past_the_operand(precision, forced_plans):
    """A K run SHORTER than one ring stage (8 channels = 16 bytes per pixel against 64-byte stages: the data gradient of a
    convolution with 8 output channels, or a convolution on an 8-channel input): the chunks past the K run must be masked from the
    FIRST stage on.  Round 5's buffer form of the tiled kernel chose in-range / out-of-range per tap but forgot the K limit in the
    initial choice: those lanes then read the following pixels -- times the zero K padding of the weight pack, harmless -- and, at the
    end of the tensor, whatever lies behind it: NaN bits of a recycled block made the BatchNorm sums of a whole layer NaN.  Here the
    operand sits at the end of a buffer whose tail is NaN: the launch must give the same bits as with a zero tail."""
    from lighthand_amd.module import HipModule

    class Net(HipModule):
        def __init__(self):
            super().__init__()
            self.a = nn.Conv2d(8, 32, 3, 1, 1, bias=False)        # forward: K run = 8 per tap
            self.b = nn.Conv2d(32, 8, 1, bias=False)              # data gradient: K run = 8

        def describe(self, gb):
            gb.output(gb.conv(gb.conv(gb.input_act(8), "a", 3, 1, 1), "b", 1, 1, 0))

    forced_plans.force_cfg = lambda cands: next(c for c in cands if c[2] not in (1, 100))      # the tiled kernel
    torch.manual_seed(5)
    m = Net().cuda().set_precision(precision).train()
    n, h, w = 3, 12, 20
    plan = m.plan(n, h, w, training=True, backward=True)
    lib, ig = plan.lib, plan._IG
    s = torch.cuda.current_stream().cuda_stream
    plan.in_act.buf.copy_(torch.randn(n, h, w, 8).to(plan.tdtype))
    plan.refresh_packs(s)
    plan.run_forward(s)
    plan.dout_nchw.copy_(torch.randn_like(plan.out_nchw))
    plan.run_backward(s)
    torch.cuda.synchronize()
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")

    def d2d(dst_ptr, src_ptr, nbytes):
        assert hip.hipMemcpy(ctypes.c_void_p(dst_ptr), ctypes.c_void_p(src_ptr), ctypes.c_size_t(nbytes), 3) == 0      # hipMemcpyDeviceToDevice

    checked = 0
    es = 2
    for lst in (plan.fwd, plan.bwd):
        for c in lst:
            if getattr(c, "fn", None) is not lib.lh_igemm or c.keep.k_run != 8:
                continue
            d = c.keep
            numel = d.n * d.hi * d.wi * d.in_pix_stride
            dst_elems = d.n * d.OH * d.OW * d.out_pix_stride
            # the launch's own operand, copied to the HEAD of a buffer whose tail is zero / NaN: the tail lies right behind the operand
            tmp = torch.empty(numel + 4096, dtype=plan.tdtype, device="cuda")
            outs, old = [], c.args
            for tail in (0.0, float("nan")):
                tmp.fill_(tail)
                torch.cuda.synchronize()
                d2d(tmp.data_ptr(), old[ig["src"]], numel * es)
                plan._patch(c, src=tmp.data_ptr())
                c(s)
                torch.cuda.synchronize()
                c.args = old
                got = torch.empty(dst_elems, dtype=plan.tdtype, device="cuda")
                d2d(got.data_ptr(), old[ig["dst"]], dst_elems * es)
                outs.append(got)
            assert not torch.isnan(outs[1].float()).any(), c.what
            assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), c.what
            checked += 1
    assert checked >= 2, checked


@pytest.mark.parametrize("shape", [(3, 64, 34, 30), (2, 128, 17, 23), (1, 64, 8, 8), (2, 64, 9, 16)])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_gated_pool_backward_equals_plain_backward_then_gate(shape, dtype):
    """lh_maxpool3x3s2_bwd_gated (the training stem: the pool follows bn1 + relu, pose_resnet.py:153-156; a 2 x 2 block of input pixels per
    thread since round 5): its dx must equal lh_maxpool3x3s2_bwd's, gated by the activation's sign, BIT FOR BIT (same additions in the same
    window order) -- even and odd sizes, ties (post-ReLU zeros) -- and its partial-sum rows must add up to sum g and sum g * xhat."""
    from lighthand_amd import _lib
    lib = _lib.load()
    n, c, h, w = shape
    td = {"bf16": torch.bfloat16, "fp16": torch.float16}[dtype]
    dt = {"bf16": _lib.LH_BF16, "fp16": _lib.LH_F16}[dtype]
    torch.manual_seed(9)
    raw = torch.randn(n, h, w, c).to(td).cuda()
    scale, shift = (0.5 + torch.rand(c)).cuda(), (0.3 * torch.randn(c)).cuda()
    mean, invstd = (0.1 * torch.randn(c)).cuda(), (0.5 + torch.rand(c)).cuda()
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    s = torch.cuda.current_stream().cuda_stream
    out, idx = torch.empty(n, ho, wo, c, dtype=td, device="cuda"), torch.empty(n, ho, wo, c, dtype=torch.uint8, device="cuda")
    _lib.check(lib.lh_bn_relu_maxpool3x3s2_fwd(raw.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(), idx.data_ptr(), n, h, w, c, dt, s))
    dy = torch.randn(n, ho, wo, c).to(td).cuda()
    plain = torch.empty(n, h, w, c, dtype=td, device="cuda")
    _lib.check(lib.lh_maxpool3x3s2_bwd(dy.data_ptr(), idx.data_ptr(), plain.data_ptr(), n, h, w, c, dt, s))
    rows = lib.lh_maxpool3x3s2_bwd_gated_rows(n, h, w, c, dt)
    partial = torch.zeros(rows, 2, c, device="cuda")
    gated = torch.full((n, h, w, c), float("nan"), dtype=td, device="cuda")
    gate = _lib.BnBwdGate(raw.data_ptr(), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), partial.data_ptr())
    _lib.check(lib.lh_maxpool3x3s2_bwd_gated(dy.data_ptr(), idx.data_ptr(), gated.data_ptr(), C.byref(gate), n, h, w, c, dt, s))
    torch.cuda.synchronize()
    on = (raw.float() * scale + shift) > 0
    want = torch.where(on, plain, torch.zeros_like(plain))
    assert torch.equal(gated, want)
    g64 = want.double()
    xhat = (raw.double() - mean.double()) * invstd.double()
    got = partial.double().sum(0)
    assert torch.allclose(got[0], g64.sum((0, 1, 2)), rtol=2e-3, atol=2e-2)
    assert torch.allclose(got[1], (g64 * xhat).sum((0, 1, 2)), rtol=2e-3, atol=2e-2)


@pytest.mark.parametrize("kernel", ["pw", "tiled"])
@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_bn_backward_gate_on_the_pointwise_kernel_and_on_residual_tails(kernel, precision, forced_plans, monkeypatch):
    """Round 6 (verdict item 2): lh_igemm_gated on the persistent pointwise kernel, and for residual tails a = relu(BN(x) + r)
    (`out += residual; relu`, pose_resnet.py:96-97), whose sign comes from the mask bits lh_fuse_fwd stored (lh_bn_bwd_gate.mask).  Three
    blocks of 1x1 convolutions, the first with a projection shortcut (two BatchNorm terms under the ReLU: lh_bn_bwd_gate.x2), two
    identity blocks: the tail of the second has two consumers -- the second block's first convolution, whose data gradient
    is the FIRST writer of its gradient and takes the second tail's identity gradient as a masked addend, and that tail itself.  With the gate
    that data gradient stores g = dz * mask and the partial sums of BN-backward; the tail's backward runs without its reduce pass.  Gate on /
    off: the gated launches appear (the tail's AND the single-term nodes' in front of a 1x1 data gradient), every gradient agrees to the
    rounding of the 16-bit activation gradients, and with PyTorch within the precision's tolerance."""
    import copy
    from lighthand_amd import _lib
    from lighthand_amd.module import HipModule
    C4, Cq, n, h, w = 128, 64, 3, 12, 20

    class Net(HipModule):
        def __init__(self):
            super().__init__()
            self.c0 = nn.Conv2d(16, C4, 1, bias=False); self.n0 = nn.BatchNorm2d(C4)
            for k in (0, 1, 2):
                setattr(self, f"a{k}", nn.Conv2d(C4, Cq, 1, bias=False)); setattr(self, f"an{k}", nn.BatchNorm2d(Cq))
                setattr(self, f"b{k}", nn.Conv2d(Cq, C4, 1, bias=False)); setattr(self, f"bn{k}", nn.BatchNorm2d(C4))
            self.p = nn.Conv2d(C4, C4, 1, bias=False); self.pn = nn.BatchNorm2d(C4)       # block 0: projection shortcut (pose_resnet.py:93-94)
            self.out = nn.Conv2d(C4, 8, 1, bias=False)

        def describe(self, gb):
            z = gb.fuse([(gb.conv(gb.input_act(16), "c0", 1, 1, 0), "n0")])
            for k in (0, 1, 2):
                r = (gb.conv(z, "p", 1, 1, 0), "pn") if k == 0 else z
                a = gb.fuse([(gb.conv(z, f"a{k}", 1, 1, 0), f"an{k}")])
                z = gb.fuse([(gb.conv(a, f"b{k}", 1, 1, 0), f"bn{k}"), r])
            gb.output(gb.conv(z, "out", 1, 1, 0))

        def torch_forward(self, x):
            z = F.relu(self.n0(self.c0(x)))
            for k in (0, 1, 2):
                r = self.pn(self.p(z)) if k == 0 else z
                a = F.relu(getattr(self, f"an{k}")(getattr(self, f"a{k}")(z)))
                z = F.relu(getattr(self, f"bn{k}")(getattr(self, f"b{k}")(a)) + r)
            return self.out(z)

    lib = _lib.load()
    torch.manual_seed(31)
    proto = Net()
    td = {"bf16": torch.bfloat16, "fp16": torch.float16}[precision]
    with torch.no_grad():
        for p_ in proto.parameters():
            p_.copy_(p_.to(td).float())
    x = torch.randn(n, 16, h, w).to(td).float()
    if kernel == "pw":
        forced_plans.force_cfg = lambda cands: max([c for c in cands if c[2] == 1] or cands[:1], key=lambda c: (c[2] == 1, c[0], c[1]))
    else:
        forced_plans.force_cfg = lambda cands: next(c for c in cands if c[2] not in (1, 100))
    res, gated = {}, {}
    dy = torch.randn(n, 8, h, w)
    for gate in ("0", "1"):
        monkeypatch.setenv("LH_BN_GATE", gate)
        m = copy.deepcopy(proto)
        out, dx, grads = _run_plan(m, x, lambda o: dy, precision)
        plan = next(iter(m._lh_plans.values()))
        calls = [c for c in plan.bwd if getattr(c, "fn", None) is lib.lh_igemm_gated]
        gated[gate] = (len(calls), sum(1 for k in plan.keep if isinstance(k, _lib.BnBwdGate) and k.mask))
        res[gate] = (out, dx, grads)
        # every gated launch against its OWN stored output: zero where the activation was not positive, and the slab rows add up to
        # sum g and sum g * xhat of what was stored (fp64 on the host), for one and for two BatchNorm terms
        gates = [k for k in plan.keep if isinstance(k, _lib.BnBwdGate)]
        assert len(gates) == len(calls)
        hip = C.CDLL("libamdhip64.so")

        def dev(ptr, numel, dtype):
            t = torch.empty(numel, dtype=dtype, device="cuda")
            assert hip.hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), C.c_size_t(numel * t.element_size()), 3) == 0
            return t
        for c, g in zip(calls, gates):
            d = c.keep
            M, Cc = d.n * d.OH * d.OW, d.cout
            assert d.out_pix_stride == Cc
            stored = dev(c.args[3], M * Cc, plan.tdtype).view(M, Cc).double()
            xin = dev(g.x, M * Cc, plan.tdtype).view(M, Cc).double()
            mean, inv = dev(g.mean, Cc, torch.float32).double(), dev(g.invstd, Cc, torch.float32).double()
            if g.mask:
                bits = dev(g.mask, M * Cc // 8, torch.uint8).view(M, Cc // 8, 1)
                on = ((bits >> torch.arange(8, device="cuda", dtype=torch.uint8).view(1, 1, 8)) & 1).bool().view(M, Cc)
            else:
                on = (xin.float() * dev(g.scale, Cc, torch.float32) + dev(g.shift, Cc, torch.float32)) > 0
            assert float(stored[~on].abs().max() if (~on).any() else 0.0) == 0.0, c.what
            rows = lib.lh_igemm_gated_rows(C.byref(d), plan.dt, 2 if g.x2 else 1)
            slabs = [(g.partial, xin, mean, inv)]
            if g.x2:
                slabs.append((g.partial2, dev(g.x2, M * Cc, plan.tdtype).view(M, Cc).double(), dev(g.mean2, Cc, torch.float32).double(),
                              dev(g.invstd2, Cc, torch.float32).double()))
            for ptr, xv, mn, iv in slabs:
                got = dev(ptr, rows * 2 * Cc, torch.float32).view(rows, 2, Cc).double().sum(0)
                want0, want1 = stored.sum(0), (stored * (xv - mn) * iv).sum(0)
                assert torch.allclose(got[0], want0, rtol=2e-3, atol=2e-3 * float(stored.abs().sum(0).max())), c.what
                assert torch.allclose(got[1], want1, rtol=2e-3, atol=2e-3 * float((stored * (xv - mn) * iv).abs().sum(0).max())), c.what
    assert gated["0"] == (0, 0), gated
    # gated: the tails (mask bits) -- block 0's has a projection shortcut (two BatchNorm terms: the pointwise kernel only), block 1's the
    # next tail as its second consumer, block 2's the output convolution as its only one -- and the single-term nodes an0 / an1 / an2
    # (their only consumer is a 1x1 data gradient)
    assert gated["1"] == ((6, 3) if kernel == "pw" else (5, 2)), gated
    a, b = res["0"], res["1"]
    assert torch.equal(a[0], b[0])
    assert rel_err(a[1], b[1]) < 2e-2, rel_err(a[1], b[1])
    for k in a[2]:
        assert rel_err(a[2][k], b[2][k]) < 3e-2, (k, rel_err(a[2][k], b[2][k]))
    ref = copy.deepcopy(proto).train()
    xr = x.clone().requires_grad_(True)
    o = ref.torch_forward(xr)
    o.backward(dy)
    assert rel_err(b[0], o.detach()) < TOL[precision]
    # a stack of four train-mode BatchNorms on random weights amplifies the 16-bit rounding of ANY path (the ungated one included) to tens
    # of percent in dx: the gated path must not be further from PyTorch than the ungated one
    e_off, e_on = rel_err(a[1], xr.grad), rel_err(b[1], xr.grad)
    print(f"{kernel} {precision}: gated launches {gated['1']}, dx vs PyTorch: gate off {e_off:.3e}, on {e_on:.3e}; on vs off {rel_err(a[1], b[1]):.3e}")
    assert e_on < 1.25 * e_off + 1e-2, (e_off, e_on)
