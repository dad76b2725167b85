"""Inference throughput of R50 256x256 bs 64 bf16 against the number of batches in flight (runtime.InferPipeline: one captured graph and
one stream per slot).  Round 5, one MI355X: 1: 33.3 k img/s, 2: 37.7 k, 3: 36.1 k, 4: 37.8 k, 6: 37.8 k -- two slots take all there is.
usage (GPU box): python tools/infer_in_flight.py"""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lighthand_amd.runtime import InferPipeline
model = bench.build_model(50, "bf16", 0)
model.eval()
images, _ = bench.synthetic_batch(64, 256, torch.device("cuda", 0), seed=9001)
for depth in (1, 2, 3, 4, 6):
    pipe = InferPipeline(model, 64, 256, 256, depth=depth)
    for st in pipe.steps:
        st.images.copy_(images)
    for _ in range(3 * depth):
        pipe.submit()
    torch.cuda.synchronize()
    best = 0
    for rep in range(3):
        t = time.perf_counter()
        n = 100 * depth
        for _ in range(n):
            pipe.submit()
        torch.cuda.synchronize()
        best = max(best, 64 * n / (time.perf_counter() - t))
    print(depth, "in flight:", round(best, 1), "img/s", flush=True)
    del pipe
