#!/usr/bin/env python3
"""Probe: two captured R50 inference graphs on two streams vs one (how much of the chip one forward pass leaves idle)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lighthand_amd.runtime import InferStep
steps = []
for i in range(2):
    m = bench.build_model(50, "bf16", 0).eval()
    s = InferStep(m, 64, 256, 256)
    im, _ = bench.synthetic_batch(64, 256, "cuda", seed=9001 + i)
    s.images.copy_(im)
    for _ in range(3):
        s()
    steps.append(s)
torch.cuda.synchronize()
def run(n, both):
    t0 = time.perf_counter()
    st = [torch.cuda.Stream(), torch.cuda.Stream()]
    for _ in range(n):
        for k in range(2 if both else 1):
            with torch.cuda.stream(st[k]):
                steps[k]()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("one graph  : %.3f ms/batch" % run(20, False))
print("two streams: %.3f ms per PAIR of batches" % run(20, True))
