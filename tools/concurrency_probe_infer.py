#!/usr/bin/env python3
"""Probe: two captured R50 inference graphs on two streams vs one (how much of the chip one forward pass leaves idle)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lighthand_amd.runtime import InferStep
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = []
for i in range(N):
    m = bench.build_model(50, "bf16", 0).eval()
    s = InferStep(m, B, 256, 256)
    im, _ = bench.synthetic_batch(B, 256, "cuda", seed=9001 + i)
    s.images.copy_(im)
    for _ in range(3):
        s()
    steps.append(s)
torch.cuda.synchronize()
def run(n, both):
    t0 = time.perf_counter()
    st = [torch.cuda.Stream() for _ in range(N)]
    for _ in range(n):
        for k in range(N if both else 1):
            with torch.cuda.stream(st[k]):
                steps[k]()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("batch %d: one graph %.3f ms; %d graphs in flight %.3f ms per round (%d images)" % (B, run(20, False), N, run(20, True), B * N))
