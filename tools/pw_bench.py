#!/usr/bin/env python3
"""Times every configuration (tiled LDS-DMA kernel and persistent pointwise kernel) of ONE 1x1 convolution launch,
forward with or without BatchNorm statistics.  usage: pw_bench.py CIN COUT N H W [stats 0|1] [iters] [stride]
LH_LIB_PATH selects an ablation build (tools/ablate.sh)."""
import ctypes as C
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LH_AUTOTUNE"] = "0"
from lighthand_amd import _lib
from lighthand_amd.module import HipModule

cin, cout, n, h, w = map(int, sys.argv[1:6])
stats = int(sys.argv[6]) if len(sys.argv) > 6 else 0
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 20
stride = int(sys.argv[8]) if len(sys.argv) > 8 else 1


class Net(HipModule):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1, stride, 0, bias=False)
        self.bn = nn.BatchNorm2d(cout)

    def describe(self, gb):
        y = gb.conv(gb.input_act(cin), "conv", 1, stride, 0)
        gb.output(gb.fuse([(y, "bn")]) if stats else y)


lib = _lib.load()
m = Net().cuda().set_precision("bf16")
plan = m.plan(n, h, w, training=True, backward=False)
plan.in_act.buf.normal_()
st = torch.cuda.current_stream()
sp = st.cuda_stream
plan.refresh_packs(sp)
call = next(c for c in plan.fwd if getattr(c, "fn", None) is lib.lh_igemm)
d = call.keep
big = torch.empty(64 << 20, dtype=torch.float32, device="cuda")          # statistics slab large enough for any candidate
if stats:
    a = list(call.args)
    a[plan._IG["stats"]] = big.data_ptr()
    call.args = tuple(a)
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")          # > Infinity Cache: every timed launch starts cold
buf = (C.c_int * (5 * 128))()
nc = lib.lh_igemm_candidates(C.byref(d), plan.dt, buf, 128)
cands = [tuple(buf[5 * i:5 * i + 4]) for i in range(nc)]
ho, wo = plan.out_act.h, plan.out_act.w
flops = 2.0 * n * ho * wo * cin * cout
nbytes = (n * h * w * cin + n * ho * wo * cout) * 2
res = []
for cfg in cands:
    for i in range(4):
        d.cfg[i] = cfg[i]
    call(sp)
    tw = tc = 0.0
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        flush.zero_()
        a.record(st); call(sp); b.record(st)
        torch.cuda.synchronize()
        tc += a.elapsed_time(b)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(iters):
        call(sp)
    b.record(st)
    torch.cuda.synchronize()
    tw = a.elapsed_time(b)
    res.append((tc / iters * 1e3, tw / iters * 1e3, cfg))
res.sort()
print(f"1x1 {cin}->{cout} s{stride} on {n}x{h}x{w} stats={stats}: {flops / 1e9:.2f} GFLOP, {nbytes / 1e6:.1f} MB (roof {nbytes / 6.3e6:.1f} us @6.3 TB/s)  lib={os.environ.get('LH_LIB_PATH', 'default')}")
for tcold, twarm, cfg in res:
    kind = "pw  " if cfg[2] == 1 else "ring"
    print(f"   {kind} {str(cfg):22s} cold {tcold:7.1f} us   back-to-back {twarm:7.1f} us")
