#!/usr/bin/env python3
"""Which BatchNorm-backward nodes of a training plan got their reduce pass from a gated data gradient (lh_igemm_gated), which did not.
usage: gate_list.py [r50|hrnet_w32] [batch] [size]"""
import sys
import torch
sys.path.insert(0, ".")
from bench import build_model          # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "r50"
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    hw = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    model = (build_model(hrnet_width=32) if name.startswith("hrnet") else build_model(50)).train()
    plan = model.plan(b, hw, hw, training=True, backward=True)
    meta = {id(c): nb for _, c, _, _, nb in plan.profile_meta}
    for c in plan.bwd:
        w = getattr(c, "what", None)
        if w and ("fuse bwd" in w or "gate" in w):
            print(f"{meta.get(id(c), 0) / 2**20:9.1f} MiB  {w}")


main()
