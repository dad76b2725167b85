#!/usr/bin/env python3
"""Probe: do two captured HRNet training steps replayed on two streams overlap on one GPU?  (Decides whether branch-level
concurrency inside the HRNet plan could pay; the R50 step showed no gain from stream overlap.)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lighthand_amd.runtime import TrainStep

width = int(sys.argv[1]) if len(sys.argv) > 1 else 32
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = []
for i in range(2):
    m = bench.build_model(50, "bf16", width)
    s = TrainStep(m, batch, 256, 256)
    im, jo = bench.synthetic_batch(batch, 256, "cuda", seed=9001 + i)
    s.images.copy_(im); s.joints.copy_(jo)
    for _ in range(3):
        s()
    steps.append(s)
torch.cuda.synchronize()
def run(n, both):
    t0 = time.perf_counter()
    if not both:
        for _ in range(n):
            steps[0]()
    else:
        st = [torch.cuda.Stream(), torch.cuda.Stream()]
        for _ in range(n):
            for k in range(2):
                with torch.cuda.stream(st[k]):
                    steps[k]()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("one graph  : %.2f ms/step" % run(10, False))
print("two streams: %.2f ms per PAIR of steps" % run(10, True))
