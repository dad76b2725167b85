#!/usr/bin/env python3
"""Sweep every compiled-in kernel configuration (lh_igemm_candidates) over the convolution shapes of a training step and
print, per launch, the static default against the fastest configurations.  Results of all configurations are compared
bit for bit with the default's (the K-loop order does not depend on the tile).
usage: conv_sweep.py [precision] [iters] [shape-filter-substring]"""
import ctypes as C
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lighthand_amd import _lib
from lighthand_amd.module import HipModule

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
filt = sys.argv[3] if len(sys.argv) > 3 else ""

# (name, cin, cout, k, stride, n, h, w, transposed)
B = 64
SHAPES = [
    ("bench 3x3 256 @64", 256, 256, 3, 1, B, 64, 64, 0),
    ("l1 1x1 64->64 @64", 64, 64, 1, 1, B, 64, 64, 0),
    ("l1 3x3 64 @64", 64, 64, 3, 1, B, 64, 64, 0),
    ("l1 1x1 64->256 @64", 64, 256, 1, 1, B, 64, 64, 0),
    ("l1 1x1 256->64 @64", 256, 64, 1, 1, B, 64, 64, 0),
    ("l2 1x1 256->128 @64", 256, 128, 1, 1, B, 64, 64, 0),
    ("l2 3x3s2 128 @64", 128, 128, 3, 2, B, 64, 64, 0),
    ("l2 1x1s2 256->512 @64", 256, 512, 1, 2, B, 64, 64, 0),
    ("l2 1x1 128->512 @32", 128, 512, 1, 1, B, 32, 32, 0),
    ("l2 1x1 512->128 @32", 512, 128, 1, 1, B, 32, 32, 0),
    ("l2 3x3 128 @32", 128, 128, 3, 1, B, 32, 32, 0),
    ("l3 1x1 512->256 @32", 512, 256, 1, 1, B, 32, 32, 0),
    ("l3 3x3s2 256 @32", 256, 256, 3, 2, B, 32, 32, 0),
    ("l3 1x1 256->1024 @16", 256, 1024, 1, 1, B, 16, 16, 0),
    ("l3 1x1 1024->256 @16", 1024, 256, 1, 1, B, 16, 16, 0),
    ("l3 3x3 256 @16", 256, 256, 3, 1, B, 16, 16, 0),
    ("l4 1x1 1024->512 @16", 1024, 512, 1, 1, B, 16, 16, 0),
    ("l4 3x3s2 512 @16", 512, 512, 3, 2, B, 16, 16, 0),
    ("l4 1x1 512->2048 @8", 512, 2048, 1, 1, B, 8, 8, 0),
    ("l4 1x1 2048->512 @8", 2048, 512, 1, 1, B, 8, 8, 0),
    ("l4 3x3 512 @8", 512, 512, 3, 1, B, 8, 8, 0),
    ("deconv0 2048->256 @8", 2048, 256, 4, 2, B, 8, 8, 1),
    ("deconv1 256->256 @16", 256, 256, 4, 2, B, 16, 16, 1),
    ("deconv2 256->256 @32", 256, 256, 4, 2, B, 32, 32, 1),
    ("head 1x1 256->21 @64", 256, 21, 1, 1, B, 64, 64, 0),
    # HRNet-W32 branch convolutions at batch 32
    ("hr b0 3x3 32 @64", 32, 32, 3, 1, 32, 64, 64, 0),
    ("hr b1 3x3 64 @32", 64, 64, 3, 1, 32, 32, 32, 0),
    ("hr b2 3x3 128 @16", 128, 128, 3, 1, 32, 16, 16, 0),
    ("hr b3 3x3 256 @8", 256, 256, 3, 1, 32, 8, 8, 0),
    # split-K stand-ins: K / 4 with 4x the pixels = the main kernel of a 4-way split of the layer-4 / layer-3 3x3 convolutions
    ("emu l4 3x3 512 @8 split4", 128, 512, 3, 1, 4 * B, 8, 8, 0),
    ("emu l4 3x3 512 @8 split2", 256, 512, 3, 1, 2 * B, 8, 8, 0),
    ("emu l3 3x3 256 @16 split2", 128, 256, 3, 1, 2 * B, 16, 16, 0),
    ("emu l4 1x1 2048->512 @8 split4", 512, 512, 1, 1, 4 * B, 8, 8, 0),
]


if os.environ.get("LH_SWEEP_SET") == "c5":      # BASELINE.json configs[4]: R50 inference, 384 x 384, batch 256 (forward launches are what counts)
    B = 256
    SHAPES = [
        ("c5 l1 3x3 64 @96", 64, 64, 3, 1, B, 96, 96, 0),
        ("c5 l2 3x3 128 @48", 128, 128, 3, 1, B, 48, 48, 0),
        ("c5 l3 3x3 256 @24", 256, 256, 3, 1, B, 24, 24, 0),
        ("c5 l4 3x3 512 @12", 512, 512, 3, 1, B, 12, 12, 0),
        ("c5 l3 1x1 1024->256 @24", 1024, 256, 1, 1, B, 24, 24, 0),
        ("c5 l4 1x1 2048->512 @12", 2048, 512, 1, 1, B, 12, 12, 0),
        ("c5 deconv0 2048->256 @12", 2048, 256, 4, 2, B, 12, 12, 1),
        ("c5 deconv1 256->256 @24", 256, 256, 4, 2, B, 24, 24, 1),
        ("c5 deconv2 256->256 @48", 256, 256, 4, 2, B, 48, 48, 1),
    ]
if os.environ.get("LH_SWEEP_SET") == "hrsmall":  # HRNet-W32 at the size of tests/test_gpu_model.py (batch 4, 128 x 96): ragged tiles, tiny K
    B = 4
    SHAPES = [
        ("stem conv2 3x3s2 64 @64x48", 64, 64, 3, 2, B, 64, 48, 0),
        ("transition1.0 3x3 256->32 @32x24", 256, 32, 3, 1, B, 32, 24, 0),
        ("transition1.1 3x3s2 256->64 @32x24", 256, 64, 3, 2, B, 32, 24, 0),
        ("fuse 3x3s2 32->64 @32x24", 32, 64, 3, 2, B, 32, 24, 0),
        ("fuse 3x3s2 32->32 @32x24", 32, 32, 3, 2, B, 32, 24, 0),
        ("fuse 3x3s2 64->128 @16x12", 64, 128, 3, 2, B, 16, 12, 0),
        ("fuse 3x3s2 128->256 @8x6", 128, 256, 3, 2, B, 8, 6, 0),
        ("transition3 3x3s2 128->256 @8x6", 128, 256, 3, 2, B, 8, 6, 0),
        ("b0 3x3 32 @32x24", 32, 32, 3, 1, B, 32, 24, 0),
        ("b1 3x3 64 @16x12", 64, 64, 3, 1, B, 16, 12, 0),
        ("b2 3x3 128 @8x6", 128, 128, 3, 1, B, 8, 6, 0),
        ("b3 3x3 256 @4x3", 256, 256, 3, 1, B, 4, 3, 0),
        ("fuse 1x1 64->32 @16x12", 64, 32, 1, 1, B, 16, 12, 0),
        ("fuse 1x1 256->32 @4x3", 256, 32, 1, 1, B, 4, 3, 0),
        ("layer1 1x1 64->256 @32x24", 64, 256, 1, 1, B, 32, 24, 0),
        ("layer1 3x3 64 @32x24", 64, 64, 3, 1, B, 32, 24, 0),
    ]
NBEST = int(os.environ.get("LH_SWEEP_NBEST", "4"))


class Net(HipModule):
    def __init__(self, cin, cout, k, s, tr):
        super().__init__()
        self.cin, self.k, self.s, self.tr = cin, k, s, tr
        self.conv = nn.ConvTranspose2d(cin, cout, k, 2, 1, 0, bias=False) if tr else nn.Conv2d(cin, cout, k, s, k // 2, bias=False)

    def describe(self, gb):
        x = gb.input_act(self.cin)
        gb.output(gb.deconv(x, "conv", self.k) if self.tr else gb.conv(x, "conv", self.k, self.s, self.k // 2))


lib = _lib.load()
st = torch.cuda.current_stream()
sp = st.cuda_stream


def descs_of(call):
    return call.keep if isinstance(call.keep, list) else [call.keep]


def timed(call, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        call(sp)
    a.record(st)
    for _ in range(n):
        call(sp)
    b.record(st)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print(f"precision {prec}, {iters} timed launches per configuration; times in us")
grand_def = grand_best = 0.0
for name, cin, cout, k, s, n, h, w, tr in SHAPES:
    if filt and filt not in name:
        continue
    os.environ["LH_AUTOTUNE"] = "0"
    m = Net(cin, cout, k, s, tr).cuda().set_precision(prec)
    plan = m.plan(n, h, w, training=True, backward=True)
    plan.in_act.buf.normal_()
    plan.dout_nchw.normal_()
    plan.refresh_packs(sp)
    plan.run_forward(sp)
    plan.run_backward(sp)
    torch.cuda.synchronize()
    calls = [c for c in list(plan.fwd) + list(plan.bwd) if getattr(c, "fn", None) in (lib.lh_igemm, lib.lh_igemm_phases)]
    ho, wo = plan.out_act.h, plan.out_act.w
    flops = 2.0 * n * (h * w if tr else ho * wo) * cin * cout * k * k
    for c in calls:
        ds = descs_of(c)
        lead = max(ds, key=lambda d: d.ntaps)
        out_t = plan.out_act.buf if c in plan.fwd else plan.in_act.grad
        buf = (C.c_int * (5 * 320))()
        nc = lib.lh_igemm_candidates(C.byref(lead), plan.dt, buf, 320)
        cands = [tuple(buf[5 * i:5 * i + 5]) for i in range(nc)]
        for d in ds:
            for i in range(5):
                d.cfg[i] = 0
        t_def = timed(c, iters)
        ref = out_t.clone()
        cur = (C.c_int * 5)()
        lib.lh_igemm_config(C.byref(lead), plan.dt, cur)
        res = []
        for cfg in cands:
            for d in ds:
                for i in range(5):
                    d.cfg[i] = cfg[i]
            t = timed(c, iters)
            same = torch.equal(out_t, ref)
            res.append((t, cfg, same))
        for d in ds:
            for i in range(5):
                d.cfg[i] = 0
        res.sort()
        fl = flops / (4 if False else 1)
        best = res[0]
        grand_def += t_def
        grand_best += best[0]
        bad = [r for r in res if not r[2]]
        print(f"{name:24s} {c.what[:26]:26s} default {tuple(cur)} {t_def:7.1f} us ({fl / t_def / 1e6:6.0f} TF/s) | best "
              + "  ".join(f"{r[1]} {r[0]:.1f}" for r in res[:NBEST]) + (f" | worst {res[-1][1]} {res[-1][0]:.1f}" if res else "")
              + (f" | MISMATCH {[r[1] for r in bad]}" if bad else "")
              + (" | best tiled " + "  ".join(f"{r[1][:4]} {r[0]:.1f}" for r in [r for r in res if r[1][2] != 1][:1])
                 + " | pointwise " + "  ".join(f"{r[1][:4]} {r[0]:.1f}" for r in res if r[1][2] == 1) if any(r[1][2] == 1 for r in res) else ""))
    if os.environ.get("LH_SWEEP_NOWGRAD"):
        del plan, m
        continue
    # ---- weight gradient: every launch plan (tile, stage rows, ring depth, pixel splits), wgrad + fold timed together
    wcalls = [c for c in plan.bwd if getattr(c, "fn", None) == lib.lh_wgrad_fused]
    for cw in wcalls:
        d = cw.args[0]._obj
        n_out, n_in = cw.args[5], cw.args[6]
        buf = (C.c_int * (5 * 320))()
        nc = lib.lh_wgrad_candidates(C.byref(d), n_out, n_in, plan.dt, buf, 320)
        cands = [tuple(buf[5 * i:5 * i + 5]) for i in range(nc)]
        ws = torch.zeros((max([c[4] for c in cands] + [0]) + 2) << 20, dtype=torch.uint8, device="cuda")
        wa = list(cw.args)
        wa[7] = ws.data_ptr()
        cw.args = tuple(wa)
        for i in range(5, 8):
            d.cfg[i] = 0
        t_def = timed(cw, iters)
        gref = plan.grads["conv.weight"].clone()
        a, b, c_, r = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        lib.lh_wgrad_tile(C.byref(d), n_out, n_in, plan.dt, C.byref(a), C.byref(b), C.byref(c_), C.byref(r))
        res = []
        for bo, bi, enc, wgs, mib in cands:
            d.cfg[5], d.cfg[6], d.cfg[7] = bo, bi, enc
            t = timed(cw, iters)
            err = float((plan.grads["conv.weight"] - gref).abs().max() / (gref.abs().max() + 1e-20))
            res.append((t, (bo, bi, (enc >> 16) & 255, enc >> 24, enc & 0xffff), err))
        for i in range(5, 8):
            d.cfg[i] = 0
        res.sort()
        grand_def += t_def
        grand_best += res[0][0] if res else t_def
        bad = [r_ for r_ in res if r_[2] > 1e-4]
        print(f"{name:24s} {'wgrad + fold':26s} default ({a.value}, {b.value}, {r.value // 10}, {r.value % 10}, {c_.value}) {t_def:7.1f} us ({flops / t_def / 1e6:6.0f} TF/s) | best "
              + "  ".join(f"{r_[1]} {r_[0]:.1f}" for r_ in res[:4]) + (f" | worst {res[-1][1]} {res[-1][0]:.1f}" if res else "")
              + (f" | MISMATCH {[r_[1] for r_ in bad]}" if bad else "")
              + (" | tap-sharing: " + "  ".join(f"{r_[1][4]} splits {r_[0]:.1f}" for r_ in res if r_[1][2] == 1) if any(r_[1][2] == 1 for r_ in res) else ""))
    del plan, m
print(f"sum of defaults {grand_def:.0f} us, sum of best {grand_best:.0f} us")
