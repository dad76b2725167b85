"""CPU, world_size 2 over gloo: the data-parallel gradient path (bucket planning + GradSync) sums every
arena slice exactly once across ranks, in backward order, and the 1/world scale gives the average."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, compress=None, algo="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from lighthand_amd import parallel
    r, w, _ = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    # a fake 3-layer "model": arena offsets and the order in which backward finishes the gradients
    offsets, off = {}, 0
    for k, n in (("a.weight", 5000), ("a.bias", 12), ("b.weight", 70000), ("b.bias", 20), ("c.weight", 30000), ("c.bias", 4)):
        offsets[k] = (off, n, (n,))
        off += (n + 3) // 4 * 4
    marks = [(3, ["c.weight", "c.bias"]), (7, ["b.bias", "b.weight"]), (9, ["a.weight", "a.bias"])]

    class FakePlan:
        bwd_marks, arena_offsets, arena_numel = marks, offsets, off

    sync = parallel.GradSync(world, bucket_bytes=4 * 20000, compress=compress, algo=algo)
    segs = sync.segments(FakePlan)
    torch.manual_seed(100 + rank)
    flat = torch.randn(off)
    mine = flat.clone()
    executed = 0
    for lo, hi, bucket in segs:            # "run bwd[lo:hi]", then reduce the slice that became final
        assert lo == executed
        executed = hi
        if bucket is not None:
            sync.launch(flat, bucket)
    sync.wait_all()
    gathered = [torch.zeros(off) for _ in range(world)]
    dist.all_gather(gathered, mine)
    want = sum(gathered)
    if compress == "bf16":       # the sum is formed and delivered in bfloat16: equal to rounding the per-rank values first
        want = sum(g.to(torch.bfloat16) for g in gathered).float()
    ok = bool(torch.allclose(flat, want, rtol=0 if compress is None else 1e-2, atol=1e-6 if compress is None else 2e-2)) \
        and executed == marks[-1][0] and len(segs) >= 2
    avg_ok = bool(torch.allclose(flat * (1.0 / world), want / world, rtol=1e-2, atol=2e-2))
    if algo == "direct":         # one rank sums each slice, in rank order, and hands the result out: every rank holds the same bits
        theirs = [torch.zeros(off) for _ in range(world)]
        dist.all_gather(theirs, flat)
        ok = ok and all(torch.equal(t, theirs[0]) for t in theirs)
    q.put((rank, ok and avg_ok, [b for _, _, b in segs]))
    dist.destroy_process_group()


@pytest.mark.parametrize("algo", ["allreduce", "direct"])
@pytest.mark.parametrize("compress", [None, "bf16"])
def test_gradsync_gloo_world2(compress, algo):
    """algo="direct": the bucket as all-to-all + local sum in rank order + all-gather (reduce-scatter and all-gather with every peer at
    once) gives the same sums as the all-reduce."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, compress, algo)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2]                    # both ranks cut identical buckets


def test_gradsync_direct_gloo_world3_pads_ragged_buckets():
    """Three ranks: no bucket of the fake model divides by 3, so the direct exchange goes through its zero-padded send buffer."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 3, port, q, None, "direct")) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


def test_gradsync_rejects_unknown_algorithm_and_takes_direct_on_the_c_abi_communicator():
    """Round 6: the direct exchange also runs on the C-ABI communicator (lh_comm_alltoall / lh_sum_chunks / lh_comm_allgather inside the
    captured step): GradSync routes a bucket to the communicator's direct_sum_ for algo="direct", to all_reduce_sum_ otherwise."""
    from lighthand_amd import parallel
    with pytest.raises(ValueError):
        parallel.GradSync(1, algo="tree")

    class FakeComm:
        def __init__(self):
            self.calls = []

        def direct_sum_(self, t):
            self.calls.append("direct")

        def all_reduce_sum_(self, t):
            self.calls.append("allreduce")

        def close(self):
            self.calls.append("close")
    for algo in ("direct", "allreduce"):
        comm = FakeComm()
        sync = parallel.GradSync(2, algo=algo, comm=comm)
        assert sync.algo == algo and sync.comm is comm
        sync.close()
        assert comm.calls == ["close"]


def test_init_distributed_single_process_is_noop(monkeypatch):
    from lighthand_amd import parallel
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert parallel.init_distributed() == (0, 1, 0)
    assert not dist.is_initialized()


def _val_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from lighthand_amd import parallel
    from lighthand_amd.tools import train as T
    parallel.init_distributed(backend="gloo")
    stopper = T.EarlyStop(float("inf"), 0, patience=2)
    # per-rank validation sums that would rank the epochs DIFFERENTLY on the two ranks (per-rank BN statistics):
    # rank 0 sees losses 1.0, 0.8, 0.9, 0.95 ; rank 1 sees 1.0, 1.1, 0.7, 1.2  ->  joint 1.0, 0.95, 0.8, 1.075, ...
    local = {0: [1.0, 0.8, 0.9, 0.95, 0.96], 1: [1.0, 1.1, 0.7, 1.2, 1.3]}[rank]
    trace = []
    for epoch, l in enumerate(local):
        acc = torch.tensor([l * 8, 8.0, 4.0, 10.0, 19.0 * 8])
        val_loss, pck, epe = T.reduce_validation(acc)
        improved, stop = stopper.update(val_loss)
        trace.append((round(val_loss, 6), improved, stop))
        # a real loop would now enter the next epoch's bucketed all-reduce: every rank must still be here
        t = torch.ones(1)
        dist.all_reduce(t)
        assert int(t) == world
        if stop:
            break
    q.put((rank, trace))
    dist.destroy_process_group()


def test_validation_decisions_are_collective_gloo_world2():
    """Early stop / best-checkpoint decisions come from the all-reduced validation sums, so the ranks leave the epoch
    loop together (a rank leaving alone would deadlock the others' gradient all-reduce)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_val_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1]
    assert [t[0] for t in res[0]] == [1.0, 0.95, 0.8, 1.075, 1.13] and res[0][-1][2] is True


def test_first_gradient_bucket_is_final_early_in_backward():
    """Data-parallel R50: the deferred weight-gradient groups end at bucket boundaries (parallel.wgrad_group_cuts), so
    the first 32 MiB bucket (head + deconvolutions) is final -- and its all-reduce in flight -- after the first few
    layers of the backward pass, not after the 24-layer group of the single-GPU plan."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from conftest import resnet_cfg
    from lighthand_amd import parallel
    from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
    m = get_pose_net(resnet_cfg(50), True)
    sd = m.state_dict()
    convs = [k[:-len(".weight")] for k, v in sd.items() if k.endswith(".weight") and v.dim() == 4]     # forward order
    back = convs[::-1]
    layer_bytes = [sd[k + ".weight"].numel() * 4 + (sd[k + ".bias"].numel() * 4 if k + ".bias" in sd else 0) for k in back]
    n = len(back)
    single = parallel.wgrad_group_cuts(layer_bytes, 24)
    dp = parallel.wgrad_group_cuts(layer_bytes, 24, 32 << 20)
    print("single-GPU group ends", single, "| data-parallel group ends", dp, "of", n, "convolutions")
    assert single[0] == 23
    assert dp[0] < 0.25 * n and len(dp) >= 4
    # bucket plan for those group ends: marks = (position, parameter names final there), arena in parameter order
    offsets, off = {}, 0
    for k, v in m.named_parameters():
        offsets[k] = (off, v.numel(), tuple(v.shape))
        off += (v.numel() + 3) // 4 * 4
    names = list(offsets)
    owner, last_conv = {}, None   # a BN's vectors are final where its convolution's node is walked: same group
    for k in names:
        if k.rsplit(".", 1)[0] in convs:
            last_conv = k.rsplit(".", 1)[0]
        owner[k] = last_conv
    marks, start = [], 0
    for end in dp + ([n - 1] if dp[-1] != n - 1 else []):
        group = set(back[start:end + 1])
        marks.append((end + 1, [k for k in names if owner[k] in group]))
        start = end + 1
    segs = parallel.plan_buckets(marks, offsets, off, 32 << 20)
    first = segs[0]
    assert first[2] is not None and (first[2][1] - first[2][0]) * 4 >= 32 << 20
    assert first[1] <= 0.25 * n, segs
