#!/usr/bin/env python3
"""The prediction the first 8-GPU run will be judged against (round-4 verdict item 6; no multi-GPU node is reachable from
the build, so this is measured compute + modelled wire).

One GPU runs the DATA-PARALLEL form of the training step (per-segment graphs cut at the gradient buckets, the collectives
stubbed: parallel.GradSync.stub) and times every segment; the table then places each bucket's all-reduce behind the
point of the step at which the bucket is final, one after the other on the side stream, at the xGMI rates of the MI355X
guide (7 links x ~153 GB/s per GPU, point to point):

  ring       2 (N-1)/N x bytes over ONE link per direction        (what a single RCCL ring is bound by)
  direct     reduce-scatter + all-gather with all 7 peers at once: 2 x bytes/N per link   

and prints, per bucket: bytes, when it is final, both all-reduce times, when each would end, and the exposed tail = what
is left of the last collective after the backward pass (the bucket's Adam slice follows it).

usage: bucket_table.py [r50|hrnet32] [world = 8] [bucket MiB = 32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, synthetic_batch                    # noqa: E402
from lighthand_amd import parallel                                # noqa: E402
from lighthand_amd.runtime import TrainStep                       # noqa: E402

LINK_GBS = 153.0


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "r50"
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    mib = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    dev = torch.device("cuda", 0)
    if which == "r50":
        model, batch, prec = build_model(50, "bf16"), 64, "bf16"
    else:
        model, batch, prec = build_model(precision="fp16", hrnet_width=32), 32, "fp16"
    sync = parallel.GradSync(world, bucket_bytes=mib << 20)
    sync.stub = True
    step = TrainStep(model, batch, 256, 256, lr=1e-3, grad_sync=sync)
    im, j = synthetic_batch(batch, 256, dev)
    step.images.copy_(im); step.joints.copy_(j)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    segs = step.graphs
    reps, acc = 10, None
    for _ in range(reps):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(segs) + 1)]
        evs[0].record()
        for i, (g, bucket) in enumerate(segs):
            if bucket == "adam":
                sync.wait_all()
            if g is not None:
                g.replay()
            if bucket is not None and bucket != "adam":
                sync.launch(step.arena.flat_grad, bucket, after=step._update_after(bucket))
            evs[i + 1].record()
        torch.cuda.synchronize()
        t = [evs[i].elapsed_time(evs[i + 1]) for i in range(len(segs))]
        acc = t if acc is None else [a + b for a, b in zip(acc, t)]
    t = [a / reps for a in acc]
    total = sum(t)
    print(f"# {which} {prec} batch {batch}/GPU, world {world}, {mib} MiB buckets: data-parallel step on ONE GPU with the collectives stubbed = {total:.3f} ms "
          f"({len(segs)} segments); wire model: {LINK_GBS:.0f} GB/s per xGMI link")
    print(f"# {'segment':>7s} {'ms':>7s} {'final at ms':>11s} {'bucket MB':>9s} | {'ring ms':>8s} {'ends at':>8s} | {'direct ms':>9s} {'ends at':>8s}")
    now, end_ring, end_dir = 0.0, 0.0, 0.0
    for i, ((g, bucket), ms) in enumerate(zip(segs, t)):
        now += ms
        if bucket is None or bucket == "adam":
            print(f"  {i:7d} {ms:7.3f} {now:11.3f} {'-' if bucket is None else 'adam tail':>9s} |")
            continue
        nbytes = (bucket[1] - bucket[0]) * 4
        ring = 2.0 * (world - 1) / world * nbytes / (LINK_GBS * 1e9) * 1e3
        direct = 2.0 * nbytes / world / (LINK_GBS * 1e9) * 1e3
        end_ring, end_dir = max(now, end_ring) + ring, max(now, end_dir) + direct
        print(f"  {i:7d} {ms:7.3f} {now:11.3f} {nbytes / 1e6:9.1f} | {ring:8.3f} {end_ring:8.3f} | {direct:9.3f} {end_dir:8.3f}")
    if os.environ.get("LH_BUCKET_TABLE_SINGLE", "1") != "0":
        # the same model as the plain single-GPU step (one graph), in this process, for the cost of the data-parallel FORM itself
        del step
        plain = TrainStep(model, batch, 256, 256, lr=1e-3)
        plain.images.copy_(im); plain.joints.copy_(j)
        for _ in range(5):
            plain()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            plain()
        e1.record()
        torch.cuda.synchronize()
        print(f"# the plain single-GPU step (one graph) in this process: {e0.elapsed_time(e1) / 20:.3f} ms")
    print(f"# backward + per-bucket Adam end at {now:.3f} ms; the last all-reduce ends at {end_ring:.3f} (ring) / {end_dir:.3f} (direct): "
          f"exposed tail {max(0.0, end_ring - now):.3f} / {max(0.0, end_dir - now):.3f} ms before the last bucket's Adam slice "
          f"(plus RCCL's launch latency per collective, ~10-20 us each, not modelled)")


if __name__ == "__main__":
    main()
