#!/usr/bin/env python3
"""Micro-benchmark of one convolution through the engine (forward igemm, data gradient, weight gradient).
usage: conv_bench.py CIN COUT K STRIDE N H W [precision] [iters]"""
import os, sys
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lighthand_amd.module import HipModule

cin, cout, k, s, n, h, w = map(int, sys.argv[1:8])
prec = sys.argv[8] if len(sys.argv) > 8 else "bf16"
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 20

class Net(HipModule):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, s, k // 2, bias=False)
    def describe(self, gb):
        gb.output(gb.conv(gb.input_act(cin), "conv", k, s, k // 2))

if os.environ.get("LH_FORCE_CFG"):            # e.g. LH_FORCE_CFG=128,128,33,128: that configuration for every tiled launch that offers it
    from lighthand_amd.engine import Plan
    want = tuple(int(v) for v in os.environ["LH_FORCE_CFG"].split(","))
    Plan.force_cfg = lambda cands: want if want in cands else None
m = Net().cuda().set_precision(prec)
plan = m.plan(n, h, w, training=True, backward=True)
plan.in_act.buf.normal_()
plan.dout_nchw.normal_()
st = torch.cuda.current_stream(); sp = st.cuda_stream
plan.refresh_packs(sp)
calls = [plan.fwd[0]] + list(plan.bwd[1:])
ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
flops = 2.0 * n * ho * wo * cin * cout * k * k
print(f"conv {cin}->{cout} k{k} s{s} on {n}x{h}x{w} {prec}: {flops/1e9:.2f} GFLOP")
for c in calls:
    for _ in range(3): c(sp)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(iters): c(sp)
    b.record(st); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    print(f"   {getattr(c,'what','?'):28s} {ms*1e3:9.1f} us  {flops/ms/1e9:8.1f} TFLOP/s")
