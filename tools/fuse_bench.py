#!/usr/bin/env python3
"""Micro-benchmark of the BN+ReLU fuse forward / backward on one big activation (HBM-bound kernels)."""
import os, sys, torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lighthand_amd.module import HipModule
c, n, h, w = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (256, 64, 64, 64)
mode = sys.argv[5] if len(sys.argv) > 5 else "plain"
class Net(HipModule):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 1, bias=False); self.bn = nn.BatchNorm2d(c); self.out = nn.Conv2d(c, 8, 1, bias=False)
    def describe(self, gb):
        x = gb.input_act(c); y = gb.conv(x, "conv", 1, 1, 0)
        z = gb.fuse([(y, "bn")] + ([x] if mode == "residual" else []))
        gb.output(gb.conv(z, "out", 1, 1, 0))
m = Net().cuda().set_precision("bf16")
plan = m.plan(n, h, w, training=True, backward=True)
plan.in_act.buf.normal_(); plan.dout_nchw.normal_()
st = torch.cuda.current_stream(); sp = st.cuda_stream
plan.refresh_packs(sp); plan.run_forward(sp); plan.run_backward(sp)
nbytes = n * h * w * c * 2
for lst, name in ((plan.fwd, "fwd"), (plan.bwd, "bwd")):
    for call in lst:
        if "fuse" not in getattr(call, "what", ""): continue
        for _ in range(3): call(sp)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for _ in range(20): call(sp)
        b.record(st); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        print(f"{os.environ.get('LH_NO_FLAT','flat'):6s} {mode:9s} {call.what:12s} {ms*1e3:8.1f} us   {nbytes/ms/1e6:8.0f} GB/s per tensor-pass")
