#!/usr/bin/env python3
"""Grid quantisation report of one training plan: workgroups per conv-family launch vs the resident slots of the chip."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lighthand_amd import _lib
from lighthand_amd.runtime import TrainStep

model = bench.build_model(50, "bf16")
step = TrainStep(model, 64, 256, 256, use_graph=False)
plan, lib = step.plan, _lib.load()
rows = []
for which, call, name, flops, nbytes in plan.profile_meta:
    d = getattr(call, "keep", None)
    if not isinstance(d, _lib.IgemmDesc) or "wgrad" in name:
        continue
    bm, bp, ring = C.c_int(0), C.c_int(0), C.c_int(0)
    lib.lh_igemm_tile(C.byref(d), plan.dt, C.byref(bm), C.byref(bp), C.byref(ring))
    M = d.n * d.ho * d.wo
    blocks = -(-M // bp.value) * -(-d.cout // bm.value)
    lds = (ring.value % 10) * (bm.value + bp.value) * 64 if ring.value else 0
    per_cu = max(1, min(8, (160 * 1024) // max(lds, 1))) if lds else 2
    slots = 256 * per_cu
    rounds = blocks / slots
    rows.append((call.what, bm.value, bp.value, blocks, per_cu, rounds, d.ntaps * ((d.k_run * 2 + 63) // 64)))
for r in rows:
    eff = r[5] / -(-r[3] // (256 * r[4])) if r[3] else 0
    print(f"{r[0][:40]:40s} tile {r[1]:3d}x{r[2]:3d} blocks {r[3]:6d} per_cu {r[4]} rounds {r[5]:6.2f} fill {eff:5.2f} ksteps {r[6]}")
