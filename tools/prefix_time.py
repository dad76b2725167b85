#!/usr/bin/env python3
"""Where does a captured step lose time that no kernel accounts for?  Un-profiled timing of PREFIXES of the forward launch
list, each captured as a hipGraph of its own: T(k) = replay time of the weight packs + the first k launches.  The increments
T(k) - T(k-1) are the cost of launch k inside a graph, without a profiler attached -- to be compared with the kernel's duration
and the idle gaps a rocprofv3 kernel trace of the full step shows at the same position (tools/step_timeline.py).
usage: prefix_time.py [hrnet|r50] [kmax] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lighthand_amd.engine import _Marker
from lighthand_amd.runtime import TrainStep

which = sys.argv[1] if len(sys.argv) > 1 else "hrnet"
kmax = int(sys.argv[2]) if len(sys.argv) > 2 else 60
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
m = bench.build_model(50, "bf16", 32 if which == "hrnet" else 0)
b = 32 if which == "hrnet" else 64
step = TrainStep(m, b, 256, 256, lr=1e-3)
im, j = bench.synthetic_batch(b, 256, torch.device("cuda"))
step.images.copy_(im); step.joints.copy_(j)
step(); step()
torch.cuda.synchronize()
plan = step.plan
real = [i for i, c in enumerate(plan.fwd) if not isinstance(c, _Marker)]
print(f"{which}: forward list {len(plan.fwd)} entries, {len(real)} launches; markers at",
      [(i, c.kind) for i, c in enumerate(plan.fwd) if isinstance(c, _Marker)][:12])


def capture(k):
    g = torch.cuda.CUDAGraph()
    warm = torch.cuda.Stream()
    warm.wait_stream(torch.cuda.current_stream())
    for it in range(2):
        ctx = torch.cuda.stream(warm) if it == 0 else torch.cuda.graph(g)
        with ctx:
            s = torch.cuda.current_stream()
            plan.refresh_packs(s.cuda_stream, overlap=True, side_work=step._render_target)
            if plan.use_lanes:
                plan._run_lanes(plan.fwd[:k], s.cuda_stream)
            else:
                for c in plan.fwd[:k]:
                    c(s.cuda_stream)
            if getattr(plan, "_pack_stream", None) is not None:          # a prefix may end before the pack streams are joined
                s.wait_stream(plan._pack_stream)
                plan._pack_event = plan._pack_event2 = None
                plan._pack_late = None
        if it == 0:
            torch.cuda.current_stream().wait_stream(warm)
            torch.cuda.synchronize()
    return g


prev = None
for k in range(4, min(kmax, len(plan.fwd)) + 1):
    if isinstance(plan.fwd[k - 1], _Marker):
        continue
    g = capture(k)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps * 1e6
    what = getattr(plan.fwd[k - 1], "what", "?")
    print(f"k {k:4d}  T {t:9.1f} us  +{(t - prev) if prev is not None else 0.0:7.1f} us   {what}")
    prev = t
    del g
