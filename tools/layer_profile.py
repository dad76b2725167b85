#!/usr/bin/env python3
"""Per-launch timing table of one eager training step (R50 bs64 bf16 by default): for every conv-family
launch prints measured ms, the MFMA and HBM lower bounds and the ratio -- shows which layers are far
from their own roofline."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--depth", type=int, default=50)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--hrnet-width", type=int, default=0)
ap.add_argument("--all", action="store_true", help="time every launch, not only the conv-family ones")
ap.add_argument("--infer", action="store_true", help="the eval-mode inference plan (BN folded) instead of the training step")
args = ap.parse_args()
from lighthand_amd.runtime import InferStep, TrainStep
model = bench.build_model(args.depth, args.precision, args.hrnet_width)
images, joints = bench.synthetic_batch(args.batch, args.size, "cuda")
if args.infer:
    model.eval()
    step = InferStep(model, args.batch, args.size, args.size, use_graph=False)
    step.images.copy_(images)
else:
    step = TrainStep(model, args.batch, args.size, args.size, use_graph=False)
    step.images.copy_(images); step.joints.copy_(joints)
plan = step.plan
meta = {id(c): (n, f, b) for w, c, n, f, b in plan.profile_meta}
stream = torch.cuda.current_stream(); s = stream.cuda_stream
rows = []
for it in range(3):
    evs = []
    plan.refresh_packs(s)
    for which, lst in ((("fwd", plan.fwd),) if args.infer else (("fwd", plan.fwd), ("bwd", plan.bwd))):
        if which == "bwd":
            step._fwd_loss_tail(s)
        for i, call in enumerate(lst):
            m = meta.get(id(call))
            if m is None and not args.all:
                call(s); continue
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream); call(s); b.record(stream)
            evs.append((which, getattr(call, "what", "?"), m or ("-", 0, 0), a, b))
    torch.cuda.synchronize()
    rows = [(w, what, m, a.elapsed_time(b)) for w, what, m, a, b in evs]
tot = 0
print(f"{'dir':3s} {'launch':42s} {'kernel':40s} {'ms':>8s} {'mfma_ms':>8s} {'hbm_ms':>8s} {'x roof':>7s} {'TF/s':>7s} {'GB/s':>7s}")
for w, what, (name, fl, by), ms in rows:
    t_m = fl / 2.5e15 * 1e3; t_h = by / 8e12 * 1e3; roof = max(t_m, t_h, 1e-9)
    tot += ms
    print(f"{w:3s} {what[:42]:42s} {name[:40]:40s} {ms:8.4f} {t_m:8.4f} {t_h:8.4f} {ms/roof:7.1f} {fl/ms/1e9 if ms else 0:7.1f} {by/ms/1e6 if ms else 0:7.0f}")
print("total profiled ms", tot)
