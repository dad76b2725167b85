"""hipGraph-captured training / inference steps: the fast path of the engine.

The reference's hot loop (src/utils/method.py:160-183) is, per iteration: H2D, forward,
JointsMSELoss, ``.item()``, full-heatmap D2H + NumPy arg-max, zero_grad, backward, Adam.
``TrainStep`` runs the same work as ONE replay of a captured HIP graph over static buffers:
weight packs -> forward -> Gaussian target render (from joints) -> MSE loss + dL/dheatmap ->
arg-max decode (kept on the device) -> backward -> [gradient all-reduce] -> fused Adam.
Loss and keypoints stay on the device; ``.loss`` / ``.preds`` are read only when the caller
asks (no per-iteration stream sync).
"""
import os

import torch

from . import _lib, heatmap
from ._lib import LightHandError, check
from .optim import Adam


def sample_color_jitter(n, brightness=0.5, contrast=0.5, saturation=0.5, hue=0.5, mask=None, generator=None):
    """torchvision ColorJitter.get_params per image (host side): factor ~ U[max(0, 1 - v), 1 + v] for brightness /
    contrast / saturation, hue ~ U[-h, h], op order = randperm(4).  ``mask`` (bool [n], optional) selects the samples
    that are jittered at all; the others get order -1 (skip).  The reference jitters a FIXED subset of its dataset, the
    samples with idx < len(meta) * ratio_of_aug (src/tools/dataset.py:133): the loader computes that flag per sample
    (lighthand_amd.tools.train) and hands it in here -- it is not a per-batch random draw.
    Returns (factors fp32 [n][4], order int32 [n][4]) CPU tensors for Plan.jitter_factors / jitter_order."""
    g = generator
    u = torch.rand(n, 4, generator=g)
    lo = torch.tensor([max(0.0, 1 - brightness), max(0.0, 1 - contrast), max(0.0, 1 - saturation), -hue])
    hi = torch.tensor([1 + brightness, 1 + contrast, 1 + saturation, hue])
    factors = (lo + (hi - lo) * u).to(torch.float32)
    order = torch.stack([torch.randperm(4, generator=g) for _ in range(n)]).to(torch.int32)
    if mask is not None:
        order[~torch.as_tensor(mask, dtype=torch.bool).cpu()] = -1
    return factors, order


def _ptr(t):
    return None if t is None else t.data_ptr()


class TrainStep:
    def __init__(self, model, batch, height, width, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 optimizer=None, decode=True, use_graph=True, grad_sync=None, targets_from_joints=True,
                 input_u8=None, color_jitter=None, loss_scale=None):
        self.lib = _lib.load()
        self.model = model
        self._model_generation = getattr(model, "_lh_generation", 0)
        model.train()
        # uint8 input rewires the plan's image launch: such a step owns its plan (model(x) in train mode keeps the float one)
        owner = ("train", "u8") if input_u8 else None
        if grad_sync is not None:               # data parallel: one set of measured kernel choices for all ranks
            from . import parallel
            self.plan = parallel.plan_with_shared_tuning(
                lambda: model.plan(batch, height, width, training=True, backward=True, wgrad_bucket_bytes=grad_sync.bucket_bytes, owner=owner))
        else:
            self.plan = model.plan(batch, height, width, training=True, backward=True, owner=owner)
        self.arena = model.arena()
        dev = self.arena.device
        out = self.plan.out_nchw
        self.images = self.plan.img_nchw                               # static input: fp32 NCHW
        # input_u8=(hs, ws): feed raw uint8 HWC frames instead; ToTensor/Resize/Normalize run fused on the device
        # color_jitter=(brightness, contrast, saturation, hue): torchvision ColorJitter ranges (reference: 0.5 each,
        # src/tools/dataset.py:139-141) applied inside the fused input kernel; factors are drawn per batch on the host
        self.color_jitter = color_jitter
        self.images_u8 = self.plan.use_uint8_input(*input_u8, jitter=color_jitter is not None) if input_u8 else None
        self.joints = torch.zeros(batch, out.shape[1], 2, dtype=torch.float32, device=dev)
        self.target = torch.zeros_like(out)
        self.targets_from_joints = targets_from_joints
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.preds = torch.zeros(batch, out.shape[1], 2, dtype=torch.float32, device=dev)
        self.maxvals = torch.zeros(batch, out.shape[1], 1, dtype=torch.float32, device=dev)
        self.decode = decode
        self._mse_ws = torch.empty(self.lib.lh_mse_workspace_bytes(out.numel()), dtype=torch.uint8, device=dev)
        self._patch = heatmap._patch_on(dev)
        self.optimizer = optimizer or Adam(model.parameters(), lr=lr, betas=betas, eps=eps)
        self.optimizer.bind_arena(self.arena)
        self.grad_sync = grad_sync                                     # parallel.GradSync or None
        self.grad_scale = 1.0 if grad_sync is None else 1.0 / grad_sync.world_size
        # static loss scaling (fp16 plans: 1024 by default): the loss GRADIENT is multiplied by S where it is formed
        # (lh_mse_heatmap), every gradient of the backward pass carries S, Adam divides it out again -- the loss value, the
        # moments and the update are those of the unscaled step, but heat-map gradients of 1e-7 no longer flush to zero in
        # the 16-bit backward pass.  bf16 / fp32 need none (fp32's exponent range).
        if loss_scale is None:
            loss_scale = 1024.0 if self.plan.tdtype == torch.float16 else 1.0
        self.loss_scale = float(loss_scale)
        self._loss_scale_dev = torch.tensor([self.loss_scale], dtype=torch.float32, device=dev) if self.loss_scale != 1.0 else None
        self.grad_scale /= self.loss_scale
        # LH_ADAM_SLICES=1: Adam slice by slice under the backward pass (Adam.apply_slice) -- the parameters of a gradient
        # bucket are updated as soon as the bucket is final (data parallel: right behind its all-reduce), on a side stream.
        # Bit-identical, and measured SLOWER on one GPU (9.73-9.80 vs 9.61-9.63 ms: a 1 GB stream through HBM and the
        # Infinity Cache under the backward kernels costs more than the 0.15 ms tail it removes), hence off by default.
        self.adam_slices = os.environ.get("LH_ADAM_SLICES", "0") == "1" and hasattr(self.optimizer, "sliceable") and self.optimizer.sliceable()
        self._adam_stream = torch.cuda.Stream() if self.adam_slices and grad_sync is None else None
        self._adam_segs = None
        self.graphs = None
        self.use_graph = use_graph
        self.heat_scale = float(height // out.shape[2])               # x4 of method.py:157
        self.steps = 0

    # ---- the work of one iteration, enqueued on the current stream --------------------------------
    def _render_target(self, stream):
        out = self.plan.out_nchw
        b, j, hs = self.joints.shape[0], self.joints.shape[1], out.shape[2]
        check(self.lib.lh_gaussian_target(self.joints.data_ptr(), 2, self._patch.data_ptr(), heatmap.RADIUS,
                                          self.target.data_ptr(), b, j, hs, stream), "lh_gaussian_target")

    def _fwd_loss(self, stream):
        # the target only depends on the joints: it is rendered on the weight-pack side stream, under the stem
        rendered = self.plan.refresh_packs(stream, overlap=True, side_work=self._render_target if self.targets_from_joints else None)
        self.plan.run_forward(stream)
        self._fwd_loss_tail(stream, target_done=rendered)

    def _fwd_loss_tail(self, stream, target_done=False):
        p, out = self.plan, self.plan.out_nchw
        if self.targets_from_joints and not target_done:
            self._render_target(stream)
        aux = None
        if self.decode and torch.cuda.current_stream().cuda_stream == stream:
            # the arg-max decode only reads the heat-maps: on a side stream under the loss kernels
            aux = self._aux_stream = getattr(self, "_aux_stream", None) or torch.cuda.Stream()
            aux.wait_event(torch.cuda.current_stream().record_event())
        check(self.lib.lh_mse_heatmap(out.data_ptr(), self.target.data_ptr(), out.numel(), self.loss.data_ptr(),
                                      p.dout_nchw.data_ptr(), _ptr(self._loss_scale_dev), self._mse_ws.data_ptr(), stream), "lh_mse_heatmap")
        if self.decode:
            check(self.lib.lh_heatmap_argmax(out.data_ptr(), out.shape[0] * out.shape[1], out.shape[2], out.shape[3],
                                             self.heat_scale, self.preds.data_ptr(), self.maxvals.data_ptr(), None,
                                             aux.cuda_stream if aux is not None else stream), "lh_heatmap_argmax")
            if aux is not None:
                torch.cuda.current_stream().wait_event(aux.record_event())

    def _enqueue_all(self):
        stream = torch.cuda.current_stream().cuda_stream
        self._fwd_loss(stream)
        if not self.adam_slices:
            self.plan.run_backward(stream)
            self.optimizer.step(grad_scale=self.grad_scale)
            return
        # the backward list runs as ONE pass; where a slice of the arena has all its gradients, its update starts on the
        # side stream behind events of every stream used so far (nothing waits for it until the end of the step)
        from . import parallel
        if self._adam_segs is None:
            mb = float(os.environ.get("LH_ADAM_SLICE_MB", "24"))
            self._adam_segs = parallel.plan_buckets(self.plan.bwd_marks, self.plan.arena_offsets, self.plan.arena_numel, int(mb * (1 << 20)))
        main, side, opt = torch.cuda.current_stream(), self._adam_stream, self.optimizer
        opt.tick(stream)

        def update(buckets):
            def fn(events):
                for e in events:
                    side.wait_event(e)
                for b in buckets:
                    opt.apply_slice(b[0], b[1], self.grad_scale, side.cuda_stream)
            return fn
        nb = len(self.plan.bwd)
        early = [(end, b) for i, (_, end, b) in enumerate(self._adam_segs) if b is not None and end < nb and i + 1 < len(self._adam_segs)]
        tail = [b for i, (_, end, b) in enumerate(self._adam_segs) if b is not None and not (end < nb and i + 1 < len(self._adam_segs))]
        at = {}
        for end, b in early:
            at.setdefault(end, []).append(b)
        self.plan.run_backward(stream, hooks={end: update(bs) for end, bs in at.items()})
        for b in tail:
            opt.apply_slice(b[0], b[1], self.grad_scale, stream)
        if early:
            main.wait_event(side.record_event())

    def _capture(self):
        """One graph for the whole step; with gradient synchronisation through torch.distributed one graph per backward
        segment, so that each gradient bucket's all-reduce (an eager call on the side stream) starts as soon as its
        segment has been replayed; with the C-ABI communicator again one graph, collectives included."""
        warm = torch.cuda.Stream()
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):          # warm-up outside capture (allocations, lazy state)
            self._enqueue_all() if self.grad_sync is None else self._eager_synced()
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        self.graphs = []
        if self.grad_sync is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._enqueue_all()
            self.graphs.append((g, None))
            return
        if getattr(self.grad_sync, "comm", None) is not None:
            # C-ABI communicator (lh_comm_*): the bucket all-reduces are stream-ordered RCCL launches that a hipGraph can
            # hold, so the whole data-parallel step -- backward segments, the all-reduces on the side stream, Adam -- is
            # ONE captured graph (no host work between segments)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._eager_synced()
            self.graphs.append((g, None))
            return
        segs = self.grad_sync.segments(self.plan)
        for i, (lo, hi, bucket) in enumerate(segs):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                stream = torch.cuda.current_stream().cuda_stream
                if i == 0:
                    self._fwd_loss(stream)
                    if self.adam_slices:
                        self.optimizer.tick(stream)
                self.plan.run_backward(stream, lo, hi)
            self.graphs.append((g, bucket))
        if self.adam_slices:                   # every bucket's update rides behind its all-reduce: nothing left but the join
            self.graphs.append((None, "adam"))
            return
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.optimizer.step(grad_scale=self.grad_scale)
        self.graphs.append((g, "adam"))

    def _update_after(self, bucket):
        """Data parallel: the Adam update of a bucket's parameters, enqueued behind the bucket's all-reduce."""
        if not self.adam_slices:
            return None
        return lambda s: self.optimizer.apply_slice(bucket[0], bucket[1], self.grad_scale, s)

    def _eager_synced(self):
        stream = torch.cuda.current_stream().cuda_stream
        self._fwd_loss(stream)
        if self.adam_slices:
            self.optimizer.tick(stream)
        for lo, hi, bucket in self.grad_sync.segments(self.plan):
            self.plan.run_backward(stream, lo, hi)
            if bucket is not None:
                self.grad_sync.launch(self.arena.flat_grad, bucket, after=self._update_after(bucket))
        self.grad_sync.wait_all()
        if not self.adam_slices:
            self.optimizer.step(grad_scale=self.grad_scale)

    def _optimizer_snapshot(self):
        """Adam moments and device step counters as they are BEFORE the warm-up / capture iterations: a resumed run
        (optimizer.load_state_dict before the first step, src/tools/train.py:50) must keep them."""
        st = self.optimizer.state.get("flat")
        moments = {k: st[k].clone() for k in ("exp_avg", "exp_avg_sq")} if st and "exp_avg" in st else None
        steps = {gi: d["step"].clone() for gi, d in self.optimizer._dev.items()}
        return moments, steps

    def _optimizer_restore(self, snap):
        """Undo the warm-up iteration on the optimizer side (weights / BN buffers are restored by ``__call__``)."""
        moments, steps = snap
        st = self.optimizer.state.get("flat")
        if st and "exp_avg" in st:
            for k in ("exp_avg", "exp_avg_sq"):
                st[k].copy_(moments[k]) if moments is not None else st[k].zero_()
        for gi, d in self.optimizer._dev.items():
            d["step"].copy_(steps[gi]) if gi in steps else d["step"].zero_()

    def close(self):
        """Release what the step owns outside PyTorch (the C-ABI communicator of a data-parallel step)."""
        if self.grad_sync is not None:
            self.grad_sync.close()

    def sync_hyper(self):
        """Push lr / betas / eps changes (e.g. CosineAnnealingLR.step()) to the device-side Adam state."""
        for gi, group in enumerate(self.optimizer.param_groups):
            st = self.optimizer._dev.get(gi)
            if st is not None:
                self.optimizer._sync_hyper(st, group)

    def __call__(self, images=None, joints=None, target=None, aug=None):
        """``aug`` (bool [batch], optional; uint8 input with color_jitter only): which samples are jittered this step
        (the reference's fixed --ratio_of_aug subset, src/tools/dataset.py:133); None = all of them."""
        if getattr(self.model, "_lh_generation", 0) != self._model_generation:
            raise LightHandError("the model's parameter storages were re-created (.to() / .cuda() / .float()) after this TrainStep "
                                 "was built: its graph still trains the old buffers -- build a new TrainStep")
        if self.graphs is not None:
            self.sync_hyper()
        if images is not None:
            (self.images_u8 if images.dtype == torch.uint8 and self.images_u8 is not None else self.images).copy_(images, non_blocking=True)
        if self.color_jitter is not None and self.images_u8 is not None:
            f, o = sample_color_jitter(self.joints.shape[0], *self.color_jitter, mask=aug)
            self.plan.jitter_factors.copy_(f, non_blocking=True)
            self.plan.jitter_order.copy_(o, non_blocking=True)
        if joints is not None:
            self.joints.copy_(joints[..., :2], non_blocking=True)
        if target is not None:
            self.target.copy_(target, non_blocking=True)
        if not self.use_graph:
            self._enqueue_all() if self.grad_sync is None else self._eager_synced()
        else:
            if self.graphs is None:
                snap = self.arena.flat.clone()
                bufs = {k: v.clone() for k, v in self.model.named_buffers()}
                opt_snap = self._optimizer_snapshot()
                self._capture()
                self._optimizer_restore(opt_snap)
                self.arena.flat.copy_(snap)                       # warm-up / capture must not train
                for k, v in self.model.named_buffers():
                    v.copy_(bufs[k])
            for g, bucket in self.graphs:
                if bucket == "adam":
                    self.grad_sync.wait_all()
                if g is not None:
                    g.replay()
                if bucket is not None and bucket != "adam":
                    self.grad_sync.launch(self.arena.flat_grad, bucket, after=self._update_after(bucket))
        self.steps += 1
        return self.loss


class InferStep:
    """Eval-mode forward + arg-max decode as one captured graph (wearable_eval_2d / pred_store path,
    src/utils/argparser.py:246-281).  ``bn_train=True`` reproduces the reference quirk of running
    ``pred_store`` without ``model.eval()`` (batch statistics at evaluation time)."""

    _serial = 0

    def __init__(self, model, batch, height, width, bn_train=False, use_graph=True, input_u8=None, slot=0):
        self.lib = _lib.load()
        # a plan of its own (never the one model(x) runs): use_uint8_input rewires the plan's image launch, and a pipeline
        # slot replays asynchronously on its own stream -- neither may happen to the plan model.forward() uses
        InferStep._serial += 1
        self.plan = model.plan(batch, height, width, training=bn_train, backward=False, slot=slot,
                               owner=("infer", "u8" if input_u8 else "f32", InferStep._serial))
        out = self.plan.out_nchw
        dev = out.device
        # input_u8=(hs, ws): raw uint8 HWC frames; ToTensor / Resize / Normalize run fused on the device (dataset.py:128-159)
        self.images = self.plan.use_uint8_input(*input_u8) if input_u8 else self.plan.img_nchw
        self.heatmaps = out
        self.preds = torch.zeros(batch, out.shape[1], 2, dtype=torch.float32, device=dev)
        self.maxvals = torch.zeros(batch, out.shape[1], 1, dtype=torch.float32, device=dev)
        self.scale = float(height // out.shape[2])
        self.use_graph = use_graph
        self.graph = None
        self._packed = False

    def _enqueue(self):
        s = torch.cuda.current_stream().cuda_stream
        self.plan.run_forward(s)
        out = self.heatmaps
        check(self.lib.lh_heatmap_argmax(out.data_ptr(), out.shape[0] * out.shape[1], out.shape[2], out.shape[3], self.scale,
                                         self.preds.data_ptr(), self.maxvals.data_ptr(), None, s), "lh_heatmap_argmax")

    def refresh_weights(self):
        self.plan.refresh_packs(torch.cuda.current_stream().cuda_stream)
        self._packed = True

    def __call__(self, images=None):
        if images is not None:
            self.images.copy_(images, non_blocking=True)
        if not self._packed:
            self.refresh_weights()
        if not self.use_graph:
            self._enqueue()
            return self.preds
        if self.graph is None:
            warm = torch.cuda.Stream()
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm):
                self._enqueue()
            torch.cuda.current_stream().wait_stream(warm)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._enqueue()
        self.graph.replay()
        return self.preds


class InferPipeline:
    """``depth`` batches in flight: one InferStep (own activation buffers, weight packs and captured graph) per slot, each on
    a stream of its own.  The stage 3-4 launches of one batch are latency-bound chains of one wave of tiles; a second batch
    fills the machine under them: R50 256x256 bs 64 bf16, 29.9 k img/s with one batch in flight, 33.3 k with two (MI355X).
    Eval-mode only (the batch-statistics quirk of ``pred_store`` updates the running statistics, which slots would race on).

        pipe = InferPipeline(model, 64, 256, 256, depth=2)
        t0 = pipe.submit(images0); t1 = pipe.submit(images1)
        preds0, maxvals0 = pipe.result(t0)          # valid until `depth` more batches have been submitted
    """

    def __init__(self, model, batch, height, width, depth=2, input_u8=None):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.steps = [InferStep(model, batch, height, width, bn_train=False, input_u8=input_u8, slot=i) for i in range(depth)]
        self.streams = [torch.cuda.Stream() for _ in range(depth)]
        self.events = [None] * depth
        self.count = 0

    def refresh_weights(self):
        """After the model's weights changed: every slot rebuilds its packs at its next submit."""
        for s in self.steps:
            s._packed = False

    def submit(self, images=None):
        i = self.count % len(self.steps)
        st = self.streams[i]
        st.wait_stream(torch.cuda.current_stream())          # the caller's copy of `images` into place is ordered before
        if images is not None and images.is_cuda:
            # the slot's copy of `images` runs on `st`: tell the caching allocator, or a caller that drops the batch right
            # after submit() gets the same block back for the next batch while this copy is still queued
            images.record_stream(st)
        with torch.cuda.stream(st):
            self.steps[i](images)
            self.events[i] = st.record_event()
        self.count += 1
        return self.count - 1

    def result(self, ticket):
        if ticket < self.count - len(self.steps) or ticket >= self.count:
            raise LightHandError(f"ticket {ticket} is not in flight (submitted so far: {self.count}, depth {len(self.steps)})")
        i = ticket % len(self.steps)
        torch.cuda.current_stream().wait_event(self.events[i])
        return self.steps[i].preds, self.steps[i].maxvals

    def map(self, batches):
        """Yields (preds, maxvals) clones for every batch of the iterable, in order, keeping `depth` batches in flight."""
        pending = []
        for images in batches:
            pending.append(self.submit(images))
            if len(pending) == len(self.steps):
                p, m = self.result(pending.pop(0))
                yield p.clone(), m.clone()
        for t in pending:
            p, m = self.result(t)
            yield p.clone(), m.clone()

