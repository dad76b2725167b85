"""Fused Adam on the HIP device (torch.optim.Adam semantics: src/tools/train.py:45-48).

A ``torch.optim.Optimizer`` subclass, so ``CosineAnnealingLR`` and ``param_groups['lr']``
work as in the reference loop.  When every parameter lives in one ``ParamArena`` the whole
model is updated by ONE launch over the flat fp32 arena; otherwise one launch per tensor.
Step count and bias corrections live on the device, so a captured hipGraph replays correctly.
"""
import torch

from . import _lib
from ._lib import check


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if weight_decay != 0:
            raise ValueError("weight_decay is not used on this path (reference: train.py:45-48 passes none)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0))
        self._flat = None        # (param_flat, grad_flat) when bound to an arena
        self._dev = {}

    def bind_arena(self, arena):
        """Use the model's flat arena: one launch per step for all parameters."""
        self._flat = arena
        return self

    # -- checkpoints interoperate with torch.optim.Adam (the reference stores optimizer.state_dict(),
    #    src/tools/dataset.py:352-360): per-parameter {step, exp_avg, exp_avg_sq} entries.
    def state_dict(self):
        sd = super().state_dict()
        flat = self.state.get("flat")
        if self._flat is None or not flat:
            return sd
        step = float(self._dev[0]["step"].item()) if 0 in self._dev else 0.0
        state, idx = {}, 0
        for group in self.param_groups:
            for p in group["params"]:
                o, n, shape = self._flat.by_param[id(p)]
                state[idx] = {"step": torch.tensor(step), "exp_avg": flat["exp_avg"][o:o + n].view(shape).clone(),
                              "exp_avg_sq": flat["exp_avg_sq"][o:o + n].view(shape).clone()}
                idx += 1
        sd["state"] = state
        return sd

    def load_state_dict(self, state_dict):
        if self._flat is None or not state_dict.get("state") or "flat" in state_dict["state"]:
            return super().load_state_dict(state_dict)
        arena = self._flat
        flat = self.state.setdefault("flat", {})
        flat.setdefault("exp_avg", torch.zeros_like(arena.flat))
        flat.setdefault("exp_avg_sq", torch.zeros_like(arena.flat))
        params = [p for g in self.param_groups for p in g["params"]]
        step = 0
        for idx, st in state_dict["state"].items():
            o, n, shape = arena.by_param[id(params[int(idx)])]
            flat["exp_avg"][o:o + n].copy_(st["exp_avg"].reshape(-1))
            flat["exp_avg_sq"][o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            step = int(float(st["step"]))
        for g, saved in zip(self.param_groups, state_dict["param_groups"]):
            g["lr"], g["betas"], g["eps"] = saved["lr"], tuple(saved["betas"]), saved["eps"]
        st = self._group_state(0, arena.device, 0)
        st["step"].fill_(step)

    def _group_state(self, gi, device, numel):
        st = self._dev.get(gi)
        if st is None:
            st = dict(hyper=torch.zeros(4, dtype=torch.float64, device=device),
                      step=torch.zeros(1, dtype=torch.int32, device=device),
                      derived=torch.zeros(8, dtype=torch.float32, device=device),
                      host=None)
            self._dev[gi] = st
        return st

    def _sync_hyper(self, st, group):
        host = (group["lr"], group["betas"][0], group["betas"][1], group["eps"])
        if st["host"] != host:
            st["hyper"].copy_(torch.tensor(host, dtype=torch.float64))
            st["host"] = host

    # -- the arena update in slices (runtime.TrainStep: a gradient bucket's parameters are updated as soon as the bucket is
    #    final, under the rest of the backward pass).  tick() once per iteration, then apply_slice() per bucket; elementwise,
    #    hence bit-identical to step().
    def sliceable(self):
        return self._flat is not None and len(self.param_groups) == 1

    def _flat_state(self):
        arena = self._flat
        st = self._group_state(0, arena.device, 0)
        s = self.state.setdefault("flat", {})
        if "exp_avg" not in s:
            s["exp_avg"] = torch.zeros_like(arena.flat)
            s["exp_avg_sq"] = torch.zeros_like(arena.flat)
        return arena, st, s

    @torch.no_grad()
    def tick(self, stream):
        arena, st, s = self._flat_state()
        self._sync_hyper(st, self.param_groups[0])
        check(_lib.load().lh_adam_tick(st["hyper"].data_ptr(), st["step"].data_ptr(), st["derived"].data_ptr(), stream), "lh_adam_tick")

    @torch.no_grad()
    def apply_slice(self, start, stop, grad_scale, stream):
        arena, st, s = self._flat_state()
        if not (0 <= start < stop <= arena.numel) or start % 4:
            raise _lib.LightHandError(f"Adam.apply_slice: bad slice [{start}, {stop}) of {arena.numel}")
        off = 4 * start
        check(_lib.load().lh_adam_apply(arena.flat.data_ptr() + off, arena.flat_grad.data_ptr() + off, s["exp_avg"].data_ptr() + off,
                                        s["exp_avg_sq"].data_ptr() + off, stop - start, st["derived"].data_ptr(), float(grad_scale), stream),
              "lh_adam_apply")

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        for gi, group in enumerate(self.param_groups):
            arena = self._flat
            if arena is not None and gi == 0 and len(self.param_groups) == 1:
                # gradients live in the arena (the engine writes them there whether or not .grad is attached)
                st = self._group_state(gi, arena.device, 0)
                self._sync_hyper(st, group)
                s = self.state.setdefault("flat", {})
                if "exp_avg" not in s:
                    s["exp_avg"] = torch.zeros_like(arena.flat)
                    s["exp_avg_sq"] = torch.zeros_like(arena.flat)
                check(lib.lh_adam_step(arena.flat.data_ptr(), arena.flat_grad.data_ptr(), s["exp_avg"].data_ptr(),
                                       s["exp_avg_sq"].data_ptr(), arena.numel, st["hyper"].data_ptr(), st["step"].data_ptr(),
                                       st["derived"].data_ptr(), float(grad_scale), stream), "lh_adam_step")
                continue
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            dev = params[0].device
            if dev.type != "cuda":
                raise _lib.LightHandError("lighthand_amd.optim.Adam runs on the HIP device only")
            st = self._group_state(gi, dev, 0)
            self._sync_hyper(st, group)
            # generic path: per-tensor launches sharing one device step counter per group.
            first = True
            for p in params:
                s = self.state[p]
                if "exp_avg" not in s:
                    s["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    s["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    s["step_t"] = torch.zeros(1, dtype=torch.int32, device=dev)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                check(lib.lh_adam_step(p.data_ptr(), g.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr(),
                                       p.numel(), st["hyper"].data_ptr(), s["step_t"].data_ptr(), st["derived"].data_ptr(),
                                       float(grad_scale), stream), "lh_adam_step")
                first = False
        return loss
