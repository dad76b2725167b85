#!/usr/bin/env python3
"""Experiment: give the deferred weight-gradient streams CUs of their own.
Eager R50 training steps with (a) the plan's default streams, (b) the weight-gradient side streams created with
hipExtStreamCreateWithCUMask on W CUs and the main stream on the other 256 - W.  usage: cumask_probe.py [steps]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lighthand_amd.runtime import TrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits[w * 32 + b]) for w in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(s.value)


def run(step, n):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


model = bench.build_model(50, "bf16")
images, joints = bench.synthetic_batch(64, 256, "cuda")
step = TrainStep(model, 64, 256, 256, use_graph=False)
step.images.copy_(images); step.joints.copy_(joints)
print(f"eager, default streams: {run(step, steps):.3f} ms/step")
plan = step.plan
default = dict(plan._lane_streams)
for layout in ("low", "strided"):
    for wcus in (32, 64, 96):
        if layout == "low":
            wbits = [i < wcus for i in range(256)]
        else:
            per = 256 // wcus                         # every per-th CU
            wbits = [i % per == 0 and i // per < wcus for i in range(256)]
        mbits = [not b for b in wbits]
        try:
            ws = [masked_stream(wbits) for _ in range(2)]
            ms = masked_stream(mbits)
        except AssertionError as e:
            print(layout, wcus, "failed:", e)
            continue
        plan._lane_streams[-1], plan._lane_streams[-2] = ws
        with torch.cuda.stream(ms):
            t = run(step, steps)
        print(f"eager, weight gradients on {wcus} CUs ({layout} mask bits), main on {256 - wcus}: {t:.3f} ms/step")
        # side streams masked, main stream unmasked
        t2 = run(step, steps)
        print(f"eager, weight gradients on {wcus} CUs ({layout}), main unmasked: {t2:.3f} ms/step")
    plan._lane_streams.update(default)
# does a captured graph keep the masks?
wbits = [i % 4 == 0 for i in range(256)]
ws = [masked_stream(wbits) for _ in range(2)]
ms = masked_stream([not b for b in wbits])
plan._lane_streams[-1], plan._lane_streams[-2] = ws
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g, stream=ms):
        step._enqueue_all()
    t = run(g.replay, steps)
    print(f"graph captured on masked streams (64 strided / 192): {t:.3f} ms/step")
except Exception as e:                                 # noqa: BLE001
    print("graph capture on masked streams failed:", repr(e)[:300])
plan._lane_streams.update(default)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    step._enqueue_all()
print(f"graph, default streams: {run(g2.replay, steps):.3f} ms/step")
