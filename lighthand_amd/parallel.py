"""Single-node data parallelism: one process per GPU, gradients averaged with bucketed RCCL
all-reduces over xGMI that overlap the rest of the backward pass.

The reference has no multi-GPU path at all; this is the MI355X-native addition
the north star asks for.  Semantics: every rank runs the same model on its own
minibatch, BatchNorm statistics stay per rank, gradients are SUMMED across ranks bucket by bucket
(buckets = contiguous slices of the flat gradient arena, cut in the order backward finishes
them) on a side HIP stream, and the 1/world factor is folded into the fused Adam launch.
Running statistics are not reduced (rank 0's are the ones saved, like the reference's
``is_main_process()`` checkpoint gate, src/tools/dataset.py:345).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Join the torchrun rendezvous (RANK / WORLD_SIZE / MASTER_* from the environment).
    Returns (rank, world_size, local_rank).  backend 'nccl' is RCCL on ROCm."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("LH_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local = local % max(torch.cuda.device_count(), 1)      # rehearsals may put several ranks on one GPU (gloo)
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if torch.cuda.is_available():
        local = local % max(torch.cuda.device_count(), 1)
    return rank, world, local


class LhComm:
    """The C-ABI communicator (lh_comm_*: RCCL underneath, include/lighthand_hip.h).  Rank 0 creates the 128-byte id,
    torch.distributed (already initialised by ``init_distributed``; any backend) carries it to the other ranks."""

    def __init__(self, rank=None, world_size=None):
        import ctypes as C
        from . import _lib
        self.lib, self._C = _lib.load(), C
        self.rank = dist.get_rank() if rank is None else rank
        self.world_size = dist.get_world_size() if world_size is None else world_size
        uid = (C.c_char * 128)()
        if self.rank == 0:
            _lib.check(self.lib.lh_comm_unique_id(uid), "lh_comm_unique_id")
        if self.world_size > 1:
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=0)
            uid = (C.c_char * 128).from_buffer_copy(box[0])
        self.handle = C.c_void_p()
        _lib.check(self.lib.lh_comm_init(C.byref(self.handle), self.rank, self.world_size, uid), "lh_comm_init")
        self._direct_bufs = {}

    def all_reduce_sum_(self, t, stream=None):
        """In-place sum of a contiguous fp32 / bf16 / fp16 device tensor on `stream` (default: the current one)."""
        from . import _lib
        assert t.is_cuda and t.is_contiguous()
        s = (stream or torch.cuda.current_stream()).cuda_stream
        _lib.check(self.lib.lh_comm_allreduce_sum(self.handle, t.data_ptr(), t.numel(), _lib.dtype_code(t.dtype), s), "lh_comm_allreduce_sum")
        return t

    def direct_sum_(self, t, stream=None):
        """In-place sum of the contiguous 1-D device tensor t over the ranks as all-to-all + local sum in RANK order (lh_sum_chunks) +
        all-gather, every launch on `stream` -- capturable, so the direct exchange lives INSIDE the single-graph step.  A length that
        divides by the rank count is exchanged in place; any other goes through a zero-padded staging buffer (allocated once per
        bucket: bucket slices of the gradient arena have stable addresses).  Every rank ends with the same bits."""
        from . import _lib
        assert t.is_cuda and t.is_contiguous() and t.dim() == 1
        w, n = self.world_size, t.numel()
        chunk = (n + w - 1) // w
        key = (t.data_ptr(), n, t.dtype)
        bufs = self._direct_bufs.get(key)
        if bufs is None:
            bufs = self._direct_bufs[key] = (t if n == w * chunk else torch.zeros(w * chunk, dtype=t.dtype, device=t.device),
                                             torch.empty(w * chunk, dtype=t.dtype, device=t.device))
        send, recv = bufs
        s = (stream or torch.cuda.current_stream()).cuda_stream
        dt = _lib.dtype_code(t.dtype)
        if send is not t:
            send[:n].copy_(t)
        _lib.check(self.lib.lh_comm_alltoall(self.handle, send.data_ptr(), recv.data_ptr(), chunk, dt, s), "lh_comm_alltoall")
        mine = send[self.rank * chunk:(self.rank + 1) * chunk]              # this rank's slice of the bucket: summed in place
        _lib.check(self.lib.lh_sum_chunks(recv.data_ptr(), mine.data_ptr(), w, chunk, dt, s), "lh_sum_chunks")
        _lib.check(self.lib.lh_comm_allgather(self.handle, mine.data_ptr(), send.data_ptr(), chunk, dt, s), "lh_comm_allgather")
        if send is not t:
            t.copy_(send[:n])
        return t

    def close(self):
        if getattr(self, "handle", None):
            self.lib.lh_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):                       # the communicator owns RCCL resources: release them with the object
        try:
            self.close()
        except Exception:                    # noqa: BLE001 -- interpreter shutdown: the library may be gone already
            pass


def all_reduce_sum_(t, group=None):
    """Sum a small tensor over the ranks in place (no-op for a single process).  Used wherever ranks must take the
    SAME host-side decision from per-rank numbers (validation loss -> best checkpoint / early stop)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def plan_with_shared_tuning(build):
    """Build a plan under torch.distributed so that EVERY rank runs the same measured kernel choices: rank 0 builds (and
    measures) first, its choices travel to the other ranks, which then build without measuring.  The weight gradient's
    pixel-split count and the forward tile fix fp32 summation orders, and a timing near-tie may fall differently per
    process: with per-rank tuning the ranks' weights would drift apart bit by bit (and every rank would pay the tuning)."""
    from .engine import Plan
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return build()
    rank = dist.get_rank()
    plan, err = None, None
    if rank == 0:
        try:
            plan = build()
        except Exception as e:                   # noqa: BLE001 -- the other ranks wait in the broadcast below: tell them
            err = e
    # NOTE: ranks != 0 wait here while rank 0 measures; a cold HRNet tune takes 1-3 minutes, well inside the process
    # group's default collective timeout (10 min for nccl) -- ship / pre-warm the tuning database for larger models
    box = [("error", repr(err)) if err is not None else ("ok", dict(Plan._TUNE_CACHE))] if rank == 0 else [None]
    dist.broadcast_object_list(box, src=0)
    if box[0][0] == "error":
        if err is not None:
            raise err
        raise RuntimeError("rank 0 failed to build the plan: " + box[0][1])
    if rank != 0:
        Plan._tune_cache_io()
        Plan._TUNE_CACHE.update(box[0][1])
        plan = build()
    return plan


def wgrad_group_cuts(layer_bytes, max_layers, bucket_bytes=None):
    """Where the backward pass hands its deferred weight-gradient groups over to the side streams.

    layer_bytes: fp32 parameter bytes of every convolution in BACKWARD order (last layer first).  A group ends after
    ``max_layers`` layers, and -- in data-parallel plans (``bucket_bytes``) -- as soon as it holds one gradient bucket's
    worth of parameters, so that the bucket's all-reduce can start while most of the backward pass is still ahead
    (the groups' ends are where ``plan_buckets`` may cut).  Returns the indices AFTER which a group is flushed."""
    cuts, n, b = [], 0, 0
    for i, nb in enumerate(layer_bytes):
        n += 1
        b += nb
        if n >= max_layers or (bucket_bytes and b >= bucket_bytes):
            cuts.append(i)
            n = b = 0
    return cuts


def plan_buckets(marks, offsets, total, bucket_bytes=32 << 20):
    """Cut the backward list into segments whose finished gradients form contiguous arena slices.

    marks:   [(end_index_in_bwd_list, [param names final at that point])] in backward order
    offsets: {name: (offset, numel, shape)} of the flat arena (fp32 elements), total = arena length
    Returns [(lo, hi, (start, stop) | None)]: run bwd[lo:hi], then all-reduce flat_grad[start:stop].
    A bucket is cut once the gradients produced so far cover a suffix [frontier, prev_cut) of at
    least ``bucket_bytes``; the last segment always flushes what is left.
    """
    order = sorted(offsets.items(), key=lambda kv: kv[1][0])
    names = [k for k, _ in order]
    starts = [v[0] for _, v in order]
    done = set()
    frontier_i = len(names)             # all params with index >= frontier_i are produced
    prev_cut = total
    segs, lo = [], 0
    for end, produced in marks:
        done.update(produced)
        while frontier_i > 0 and names[frontier_i - 1] in done:
            frontier_i -= 1
        frontier = starts[frontier_i] if frontier_i < len(names) else total
        if (prev_cut - frontier) * 4 >= bucket_bytes:
            segs.append((lo, end, (frontier, prev_cut)))
            lo, prev_cut = end, frontier
    last_end = marks[-1][0] if marks else 0
    if frontier_i != 0:
        raise RuntimeError("backward never produced gradients for: " + ", ".join(names[:frontier_i][:5]))
    if prev_cut > 0 or lo < last_end:
        segs.append((lo, last_end, (0, prev_cut) if prev_cut > 0 else None))
    return segs


class GradSync:
    """Launches one all-reduce per gradient bucket on a side stream; the optimizer waits for all of them (or updates each
    bucket's parameters right behind its all-reduce: launch(after=...))."""

    def __init__(self, world_size=None, bucket_bytes=32 << 20, group=None, compress=None, comm=None, algo="allreduce"):
        """compress='bf16': every bucket travels as bfloat16 (half the bytes per xGMI link; the sum is formed in bf16 by
        the collective, the fp32 arena slice receives the result).  Default: fp32 buckets, exact sums.
        comm: an ``LhComm`` -- the buckets then go through the C-ABI communicator (lh_comm_allreduce_sum) instead of
        torch.distributed's all_reduce (same RCCL underneath)."""
        self.comm = comm
        # algo="direct": a bucket is exchanged as all-to-all + local sum in rank order + all-gather -- reduce-scatter and all-gather
        # with ALL peers at once (2 x bytes / N per xGMI link instead of a ring's 2 (N - 1) / N x bytes over one), and
        # every rank ends with bit-identical sums.  "allreduce" leaves the algorithm to RCCL.  With the C-ABI communicator the three steps
        # are lh_comm_alltoall / lh_sum_chunks / lh_comm_allgather on the side stream (LhComm.direct_sum_): capturable, one graph per step.
        if algo not in ("allreduce", "direct"):
            raise ValueError("algo must be 'allreduce' or 'direct'")
        self.algo = algo
        self._direct_bufs = {}
        self.world_size = world_size or (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.bucket_bytes = bucket_bytes
        self.group = group
        if compress not in (None, "bf16"):
            raise ValueError("compress must be None or 'bf16'")
        self.compress = compress
        self._staging = {}
        self.cuda = torch.cuda.is_available()
        self.stream = torch.cuda.Stream() if self.cuda else None
        self._pending = []
        self._segments = None
        # measurement only (bench.py allreduce_exposed_ms): with stub = True launch() keeps its stream hand-over, the bf16
        # staging and the `after` work but skips the collective itself -- the step then costs what it would with a free wire
        self.stub = False

    def _direct_sum_(self, t):
        """In-place sum of the 1-D tensor t over the ranks: all-to-all of its world_size chunks, local sum in rank order, all-gather.
        A length that divides by world_size is exchanged in place (no staging copy); any other goes through a zero-padded send buffer."""
        w, n = self.world_size, t.numel()
        chunk = (n + w - 1) // w
        key = (t.data_ptr(), n, t.dtype) if t.is_cuda else None   # (device buckets are arena slices: stable addresses)
        bufs = self._direct_bufs.get(key) if key else None
        if bufs is None:
            bufs = (t if n == w * chunk else torch.zeros(w * chunk, dtype=t.dtype, device=t.device),
                    torch.empty(w * chunk, dtype=t.dtype, device=t.device), torch.empty(chunk, dtype=t.dtype, device=t.device))
            if key:
                self._direct_bufs[key] = bufs
        send, recv, mine = bufs
        if send is not t:
            send[:n].copy_(t)                                      # (the pad behind n stays zero)
        dist.all_to_all_single(recv, send, group=self.group)       # recv chunk r = rank r's part of the slice this rank owns
        if t.is_cuda:
            from . import _lib
            lib = _lib.load()
            _lib.check(lib.lh_sum_chunks(recv.data_ptr(), mine.data_ptr(), w, chunk, _lib.dtype_code(t.dtype),
                                         torch.cuda.current_stream().cuda_stream), "lh_sum_chunks")
        else:
            acc = recv[:chunk].float()
            for r in range(1, w):
                acc = acc + recv[r * chunk:(r + 1) * chunk].float()
            mine.copy_(acc.to(t.dtype))
        if dist.get_backend(self.group) == "gloo":
            dist.all_gather(list(send.view(w, chunk).unbind(0)), mine, group=self.group)
        else:
            dist.all_gather_into_tensor(send, mine, group=self.group)
        if send is not t:
            t.copy_(send[:n])
        return t

    def segments(self, plan):
        if self._segments is None:
            offsets = plan.arena_offsets
            self._segments = plan_buckets(plan.bwd_marks, offsets, plan.arena_numel, self.bucket_bytes)
        return self._segments

    def launch(self, flat_grad, bucket, after=None):
        """All-reduce flat_grad[start:stop] on the side stream.  after(stream): work to enqueue on that stream right behind
        the collective (TrainStep: the Adam update of the bucket's parameters); device tensors only."""
        start, stop = bucket
        view = flat_grad[start:stop]
        if after is not None and not (self.cuda and view.is_cuda):
            raise ValueError("GradSync.launch(after=...) needs device gradients")
        if self.world_size == 1:
            if after is not None:
                after(torch.cuda.current_stream().cuda_stream)
            return
        if self.cuda and view.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                reduce_ = (lambda t: None) if self.stub else \
                    (self.comm.direct_sum_ if self.algo == "direct" else self.comm.all_reduce_sum_) if self.comm is not None else \
                    self._direct_sum_ if self.algo == "direct" else (lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group))
                if self.compress == "bf16":
                    half = self._staging.get(bucket)
                    if half is None:
                        half = self._staging[bucket] = torch.empty(stop - start, dtype=torch.bfloat16, device=view.device)
                    # staging through the C ABI (lh_cast_f32_bf16): the data-parallel step holds no framework compute
                    from . import _lib
                    lib, sp = _lib.load(), self.stream.cuda_stream
                    _lib.check(lib.lh_cast_f32_bf16(view.data_ptr(), half.data_ptr(), stop - start, 0, sp), "lh_cast_f32_bf16")
                    reduce_(half)
                    _lib.check(lib.lh_cast_f32_bf16(view.data_ptr(), half.data_ptr(), stop - start, 1, sp), "lh_cast_f32_bf16")
                else:
                    reduce_(view)
                if after is not None:
                    after(self.stream.cuda_stream)
                done = torch.cuda.Event()
                done.record(self.stream)
            self._pending.append(done)
        elif self.compress == "bf16":
            half = view.to(torch.bfloat16)
            if self.algo == "direct":
                self._direct_sum_(half)
            else:
                dist.all_reduce(half, op=dist.ReduceOp.SUM, group=self.group)
            view.copy_(half)
        elif self.algo == "direct":
            self._direct_sum_(view)
        else:
            self._pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def close(self):
        """Release the C-ABI communicator, if this object drives one (TrainStep.close / interpreter exit)."""
        if self.comm is not None:
            self.comm.close()

    def wait_all(self):
        for p in self._pending:
            if isinstance(p, torch.cuda.Event):
                torch.cuda.current_stream().wait_event(p)
            else:
                p.wait()
        self._pending = []
