#!/usr/bin/env python3
"""What the DATA-PARALLEL FORM of the training step costs on ONE GPU before any wire time: the plain step against
 (a) the default transport's form -- one graph per backward segment with the collectives stubbed (GradSync.stub), host launches in between;
 (b) the one-graph form on the C-ABI communicator with a ONE-RANK RCCL communicator (real, stream-ordered collectives that move nothing):
     RCCL's all-reduce per bucket, and the direct exchange (lh_comm_alltoall + lh_sum_chunks + lh_comm_allgather).
usage (GPU box): python tools/dp_form_cost.py [bucket MiB] [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lighthand_amd import parallel
from lighthand_amd.runtime import TrainStep

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
images, joints = bench.synthetic_batch(64, 256, dev)


def run(tag, make_sync, stub=False):
    model = bench.build_model(50, "bf16")
    sync = make_sync()
    step = TrainStep(model, 64, 256, 256, lr=1e-3, grad_sync=sync)
    step.images.copy_(images); step.joints.copy_(joints)
    if stub:
        sync.stub = True
    ms = bench.timed_replays(step, 20, steps) * 1e3
    n_graphs = len(step.graphs)
    print(f"{tag:64s} {ms:7.3f} ms per step   ({n_graphs} graph(s) per step)", flush=True)
    step.close()
    return ms


comm = parallel.LhComm(rank=0, world_size=1)
for rep in range(2):
    base = run("plain step (no gradient exchange)", lambda: None)
    run(f"torch transport form, collectives stubbed, {mib} MiB buckets", lambda: parallel.GradSync(world_size=2, bucket_bytes=mib << 20), stub=True)
    run(f"one graph, lh_comm all-reduce (1-rank RCCL), {mib} MiB buckets", lambda: parallel.GradSync(world_size=2, bucket_bytes=mib << 20, comm=parallel.LhComm(rank=0, world_size=1)))
    run(f"one graph, lh_comm direct exchange (1-rank RCCL), {mib} MiB buckets", lambda: parallel.GradSync(world_size=2, bucket_bytes=mib << 20, comm=parallel.LhComm(rank=0, world_size=1), algo="direct"))
