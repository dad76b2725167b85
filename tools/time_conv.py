#!/usr/bin/env python3
"""Time ONE convolution launch (forward, through the engine) under forced kernel configurations, un-profiled.
usage: time_conv.py CIN COUT K STRIDE N H W TRANSPOSED prec "bm,bp,depth,kb;..." [iters]
With LH_LIB_PATH=tools/abl/lib_ablN.so (tools/ablate.sh) this is how the ablation tables of DESIGN.md 3.1a are made."""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LH_AUTOTUNE"] = "0"
from lighthand_amd import _lib
from lighthand_amd.module import HipModule

cin, cout, k, s, n, h, w, tr = map(int, sys.argv[1:9])
prec = sys.argv[9]
cfgs = [tuple(int(v) for v in c.split(",")) for c in sys.argv[10].split(";")]
iters = int(sys.argv[11]) if len(sys.argv) > 11 else 20


class Net(HipModule):
    def __init__(self):
        super().__init__()
        self.conv = nn.ConvTranspose2d(cin, cout, k, 2, 1, 0, bias=False) if tr else nn.Conv2d(cin, cout, k, s, k // 2, bias=False)

    def describe(self, gb):
        x = gb.input_act(cin)
        gb.output(gb.deconv(x, "conv", k) if tr else gb.conv(x, "conv", k, s, k // 2))


lib = _lib.load()
m = Net().cuda().set_precision(prec)
plan = m.plan(n, h, w, training=False, backward=False)
plan.in_act.buf.normal_()
sp = torch.cuda.current_stream().cuda_stream
plan.refresh_packs(sp)
call = [c for c in plan.fwd if getattr(c, "fn", None) in (lib.lh_igemm, lib.lh_igemm_phases)][0]
ds = call.keep if isinstance(call.keep, list) else [call.keep]
flop = 2.0 * n * (h // s) * (w // s) * cin * cout * k * k * (1 if not tr else 1)
for cfg in cfgs:
    for d in ds:
        for i in range(4):
            d.cfg[i] = cfg[i]
    for _ in range(3):
        call(sp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call(sp)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{os.environ.get('LH_LIB_PATH', 'product')} cfg {cfg}: {us:8.1f} us  {flop / us / 1e6:7.0f} TFLOP/s")
