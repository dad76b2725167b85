#!/usr/bin/env python3
"""Times lh_bottleneck_infer alone on the stage-1 shapes of BASELINE.json configs[4] (R50 inference, 384 x 384, batch 256, fp16:
256 x 96 x 96 pixels) -- an identity block (cin 256) and the block behind the projection (cin 64) -- and prints us per launch,
TFLOP/s and the HBM rate of its algorithmic bytes (input + residual + output).  LH_LIB_PATH selects an ablation build
(tools/ablate_bneck.sh).   usage: bneck_bench.py [batch = 256] [size = 96] [iters = 10]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lighthand_amd import _lib

lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 96
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
for cin in (256, 64):
    x = torch.randn(n, hw, hw, cin, device=dev, dtype=torch.float16, generator=g)
    res = torch.randn(n, hw, hw, 256, device=dev, dtype=torch.float16, generator=g)
    out = torch.empty_like(res)
    kp = (cin + 63) // 64 * 64
    w1 = (torch.randn(128, kp, device=dev, generator=g) * 0.05).half()
    w2 = (torch.randn(128, 9 * 64, device=dev, generator=g) * 0.05).half()
    w3 = (torch.randn(256, 64, device=dev, generator=g) * 0.05).half()
    sc = [torch.rand(c, device=dev) + 0.5 for c in (64, 64, 256)]
    sh = [torch.randn(c, device=dev) * 0.1 for c in (64, 64, 256)]
    d = _lib.BottleneckDesc(n, hw, hw, cin, 64, 256)
    s = torch.cuda.current_stream().cuda_stream

    def run():
        _lib.check(lib.lh_bottleneck_infer(C.byref(d), x.data_ptr(), w1.data_ptr(), w2.data_ptr(), w3.data_ptr(), sc[0].data_ptr(), sh[0].data_ptr(),
                                           sc[1].data_ptr(), sh[1].data_ptr(), sc[2].data_ptr(), sh[2].data_ptr(), res.data_ptr(), out.data_ptr(), _lib.LH_F16, s),
                   "lh_bottleneck_infer")
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    px = n * hw * hw
    flops = 2.0 * px * (cin * 64 + 9 * 64 * 64 + 64 * 256)
    nbytes = 2.0 * px * (cin + 256 + 256)
    print(f"{os.path.basename(_lib.LIB_PATH):28s} cin {cin:4d}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  {nbytes / us / 1e6:6.2f} TB/s algorithmic")
