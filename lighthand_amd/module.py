"""Host-side base class of the drop-in models: an ``nn.Module`` whose parameters keep the
reference's names/layouts (so ``state_dict`` / ``load_state_dict`` / optimizers work as in
the reference) but whose ``forward`` runs the HIP engine -- never torch.nn kernels.
"""
import torch
import torch.nn as nn

from . import _lib
from .engine import PRECISIONS, Plan


class ParamArena:
    """All trainable parameters in one flat fp32 buffer (+ one for gradients).

    Parameters become views of the flat buffer, so a single fused-Adam launch (and a single
    bucketed all-reduce per slice in data-parallel runs) covers the whole model while
    ``state_dict()`` still yields the reference's tensors.
    """

    def __init__(self, module):
        named = [(k, p) for k, p in module.named_parameters()]
        self.device = named[0][1].device
        self.offsets, off = {}, 0
        for k, p in named:
            self.offsets[k] = (off, p.numel(), tuple(p.shape))
            off += (p.numel() + 3) // 4 * 4          # keep every slice 16-byte aligned
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=self.device)
        for k, p in named:
            o, n, shape = self.offsets[k]
            view = self.flat[o:o + n].view(shape)
            view.copy_(p.data)
            p.data = view
        self.names = [k for k, _ in named]
        self.by_param = {id(p): self.offsets[k] for k, p in named}

    def grad_view(self, name):
        o, n, shape = self.offsets[name]
        return self.flat_grad[o:o + n].view(shape)


class _ModelFn(torch.autograd.Function):
    """Whole-model autograd node: forward = plan forward list, backward = plan backward list."""

    @staticmethod
    def forward(ctx, module, plan, x, *params):
        out = plan.forward(x.detach().to(torch.float32))
        plan.generation += 1
        ctx.module, ctx.plan, ctx.generation, ctx.nparams = module, plan, plan.generation, len(params)
        return out.clone()

    @staticmethod
    def backward(ctx, dheat):
        plan, module = ctx.plan, ctx.module
        if ctx.generation != plan.generation:
            raise _lib.LightHandError("backward() of a stale forward: the engine keeps activations of the most recent "
                                      "forward only (same restriction as the reference's single-stream training loop)")
        arena = module._lh_arena
        params = dict(module.named_parameters())
        carry = {}
        for k, p in params.items():           # gradients that must be accumulated, not replaced
            if p.grad is not None:
                carry[k] = p.grad.clone() if p.grad.data_ptr() == arena.grad_view(k).data_ptr() else p.grad
        plan.backward(dheat.contiguous().to(torch.float32))
        for k, p in params.items():
            g = arena.grad_view(k)
            if k in carry:
                g.add_(carry[k])
            p.grad = g
        return (None, None, None) + (None,) * ctx.nparams


class HipModule(nn.Module):
    """Shared engine plumbing for PoseResNet / PoseHighResolutionNet."""

    def __init__(self):
        super().__init__()
        self._lh_precision = "fp32"
        self._lh_plans = {}
        self._lh_arena = None
        self._lh_generation = 0          # bumped whenever the parameter storages are re-created (plans / steps built before are stale)

    # -- configuration --------------------------------------------------------------------------
    def set_precision(self, precision):
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
        if precision != self._lh_precision:
            self._lh_precision = precision
            self._lh_plans.clear()
        return self

    @property
    def precision(self):
        return self._lh_precision

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float() may re-create the parameter storages: drop arena and plans -- but only when a storage,
        # device or dtype really changed (model.cuda() on a model that already lives there is a no-op and must leave a live
        # TrainStep / optimizer, which hold raw pointers into the arena, intact)
        def sig():
            return [(t.data_ptr(), t.device, t.dtype) for t in list(self.parameters()) + list(self.buffers())]
        before = sig()
        out = super()._apply(fn, *args, **kwargs)
        if sig() != before:
            self._lh_plans.clear()
            self._lh_arena = None
            self._lh_generation = getattr(self, "_lh_generation", 0) + 1
        return out

    def __getstate__(self):
        # plans hold raw device pointers / ctypes descriptors: never copied or pickled with the module
        st = dict(self.__dict__)
        st["_lh_plans"], st["_lh_arena"] = {}, None
        return st

    def load_state_dict(self, state_dict, strict=True, **kw):
        # copies in place, so the arena views and every bound pointer stay valid
        return super().load_state_dict(state_dict, strict=strict, **kw)

    # -- engine ---------------------------------------------------------------------------------
    def describe(self, gb):
        raise NotImplementedError

    def arena(self):
        if self._lh_arena is None:
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise _lib.LightHandError("lighthand_amd models compute on a HIP device only: call .to('cuda') first "
                                          "(there is no CPU fallback)")
            _lib.load()
            self._lh_arena = ParamArena(self)
        return self._lh_arena

    def plan(self, n, h, w, training=None, backward=None, wgrad_bucket_bytes=None, slot=0, owner=None):
        """slot: plans of the same shape with different slots own separate activation buffers and weight packs (several
        batches in flight on different streams: runtime.InferPipeline).
        owner: a step object that REWIRES or replays its plan on its own (runtime.InferStep with uint8 input or on a
        pipeline stream, runtime.TrainStep) names itself here and gets a plan of its own, which the module does NOT keep
        (it lives and dies with the step): ``model(x)`` keeps the (shape, mode) plan without an owner, whose input is
        always the float NCHW image and whose buffers no asynchronous replay touches."""
        training = self.training if training is None else training
        backward = training if backward is None else backward
        self.arena()
        if owner is not None:
            # never cached: the step holds the only reference, so dropping the step frees the plan's activation buffers,
            # weight packs and workspaces (a cached owner plan per InferStep made device memory grow without bound)
            p = Plan(self, n, h, w, self._lh_precision, training=training, backward=backward, wgrad_bucket_bytes=wgrad_bucket_bytes)
            p.generation = 0
            return p
        key = (n, h, w, self._lh_precision, training, backward, wgrad_bucket_bytes) + ((slot,) if slot else ())
        p = self._lh_plans.get(key)
        if p is None:
            p = Plan(self, n, h, w, self._lh_precision, training=training, backward=backward, wgrad_bucket_bytes=wgrad_bucket_bytes)
            p.generation = 0
            self._lh_plans[key] = p
        return p

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected an NCHW image batch with 3 channels, got {tuple(x.shape)}")
        if not x.is_cuda:
            raise _lib.LightHandError("input must live on the HIP device (images.cuda() as in the reference loop)")
        n, _, h, w = x.shape
        # gradients flow only in train mode (batch-statistics BN), as in the reference's loops; an eval-mode
        # forward is inference (running statistics folded into the conv epilogues) and returns a plain tensor
        need_grad = self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        plan = self.plan(n, h, w, training=self.training, backward=need_grad)
        if need_grad:
            return _ModelFn.apply(self, plan, x, *self.parameters())
        return plan.forward(x.detach().to(torch.float32)).clone()
