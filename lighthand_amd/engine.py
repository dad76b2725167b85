"""Static-shape execution engine: turns a model's layer description into two flat lists
of pre-bound HIP launches (forward, backward) over pre-allocated NHWC buffers.

There is no tracing compiler and no per-op autograd: a model describes itself once through
``GraphBuilder`` (conv / deconv / fuse / maxpool nodes that name their parameters by the
reference's state_dict keys), ``Plan`` allocates every activation, gradient, weight pack
and workspace for one (batch, H, W, precision, mode), and running the network is a loop
of ctypes calls on the caller's HIP stream -- which is exactly what a hipGraph captures
(``lighthand_amd.runtime.TrainStep``).
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import FuseBwdDesc, FuseDesc, IgemmDesc, check
from .batch_groups import BatchGroups
from .graph import BN_EPS, BN_MOMENTUM, PRECISIONS, Act, GraphBuilder, _Call, _Marker, _desc, _ptr, _taps_array  # noqa: F401 (re-exported)
from .infer_rewrites import InferRewrites
from .tuner import Tuner
from .wgrad_schedule import WgradSchedule


# --------------------------------------------------------------------------------------- plan
class Plan(Tuner, WgradSchedule, BatchGroups, InferRewrites):
    """Everything needed to run one model at one static shape."""

    def __init__(self, model, n, h, w, precision="fp32", training=True, backward=None, device=None, wgrad_bucket_bytes=None):
        self.lib = _lib.load()
        self.precision = precision
        self.tdtype = PRECISIONS[precision]
        self.dt = _lib.dtype_code(self.tdtype)
        self.es = self.tdtype.itemsize if hasattr(self.tdtype, "itemsize") else torch.tensor([], dtype=self.tdtype).element_size()
        self.epc = 16 // self.es
        self.training = training                 # BatchNorm mode: batch statistics + running update
        self.with_bwd = training if backward is None else backward
        self.n, self.h, self.w = n, h, w
        self.params = dict(model.state_dict(keep_vars=True))
        self.device = device or next(iter(self.params.values())).device
        if self.device.type != "cuda":
            raise _lib.LightHandError("lighthand_amd runs on a HIP device only; move the model with .to('cuda')")
        self.grads = {}
        arena0 = getattr(model, "_lh_arena", None)
        self.arena_offsets = arena0.offsets if arena0 is not None else {}
        self.arena_numel = arena0.numel if arena0 is not None else 0
        if self.with_bwd:
            arena = getattr(model, "_lh_arena", None)
            for k, p in self.params.items():
                if isinstance(p, torch.nn.Parameter):
                    self.grads[k] = arena.grad_view(k) if arena is not None else torch.zeros_like(p)
        gb = GraphBuilder(n, h, w, self.params)
        model.describe(gb)
        self.nodes = gb.nodes
        self.node_lanes = gb.node_lanes
        self.node_branch, inside = [], False                   # node sits inside a fork .. join region (branch 0 runs on lane 0, and
        for kind, _ in gb.nodes:                                # _batch_order moves grouped regions to lane 0: the lanes do not say it)
            inside = (inside or kind == "fork") and kind != "join"
            self.node_branch.append(inside)
        self.n_lanes = max(gb.node_lanes) + 1 if gb.node_lanes else 1
        self.use_lanes = self.n_lanes > 1
        # Multi-problem launches: the nodes at the same position of the parallel chains of a fork .. join region (HRNet's
        # branches) are compiled as a GROUP whose launches merge into lh_*_multi calls (one grid for 2-4 problems) on the
        # main stream, instead of one launch per branch on stream lanes.  16-bit types; LH_BATCH=0 keeps the lanes.
        self.batch = os.environ.get("LH_BATCH", "1") != "0" and self.es == 2 and self.n_lanes > 1
        self.wgrad_batch = os.environ.get("LH_WGRAD_BATCH", "1") != "0" and self.es == 2
        # Table launches (round 6): ALL weight gradients of a deferred group that share a tile class run as ONE grid with a split count
        # per layer + at most one fold grid (lh_wgrad_table_run; _table_wgrads).  LH_WGRAD_TABLE=0: one launch (+ fold) per layer.
        self.wgrad_table = os.environ.get("LH_WGRAD_TABLE", "1") != "0" and self.es == 2
        self.wgrad_tables = []             # (call, info, member names) of every table launch of the plan
        self._forced = None                # kernel choices of the group being compiled (see _tune_group)
        self._n_groups = 0
        self._lane_streams = {L: torch.cuda.Stream(device=self.device) for L in range(1, self.n_lanes)} if self.use_lanes else {}
        self._cur_lane = 0
        self._emit_group = 1           # size of the batch group whose backward blocks are being emitted
        # Deferred weight gradients: the weight-gradient launches (+ folds) of a GROUP of layers are appended behind one
        # event and run on a side stream while the stream that produced their dy walks on (the main stream of a
        # single-lane network; a branch lane of HRNet: its chain inside one module).  Groups alternate over the side
        # streams, each with its own split-K slab.  LH_WGRAD_GROUP = layers per group (0 = weight gradients in place).
        n_convs = sum(1 for k, _ in self.nodes if k in ("conv", "deconv"))
        auto_group = max(4, -(-n_convs * 62 // 100))         # 36 layers for R50 (re-measured in round 3: 24: 9.65 ms, 32-40: 9.59-9.61, 48: 9.77)
        if self.n_lanes > 1:
            auto_group = 16                                   # branch lanes hand over at every module end; 16 on the main lane
        self.wgrad_group = int(os.environ.get("LH_WGRAD_GROUP", str(auto_group))) if self.with_bwd else 0
        # data-parallel plans: a deferred group ALSO ends as soon as its layers hold one gradient bucket's worth of
        # parameters, so the first bucket's all-reduce starts early in the backward pass (parallel.wgrad_group_cuts)
        self.wgrad_bucket_bytes = wgrad_bucket_bytes
        if self.wgrad_group > 0:
            self.use_lanes = True
            self._w_lanes = int(os.environ.get("LH_WGRAD_LANES", "2" if self.n_lanes == 1 else "4"))
            for i in range(self._w_lanes):
                self._lane_streams[-1 - i] = torch.cuda.Stream(device=self.device)
        self._pend = {}                    # source lane -> dict(calls, names, layers, ws)
        self._w_flushes = 0
        self._ready = {}
        self.out_act = gb.out
        self.fwd, self.bwd, self.packs = [], [], []
        self._pack_items = []
        self._pack_convs = {}              # id(weight) -> PackConv (LDS-tiled transposing pack)
        self._producers = {}               # id(raw conv output Act) -> the igemm calls that write it (eval-mode BN folding)
        self.keep = []                     # ctypes objects / tensors referenced by raw pointer
        self._ws_wgrad = 0
        self._ws_fuse = 0
        self._ws_users = []
        self._ws_users_fuse = []
        self.profile_meta = []             # (list name, call object, kernel name, flops, bytes)
        self._tune_bufs = {}
        Plan._tune_cache_io()
        n_tuned = len(Plan._tune_measured)
        self._compile()
        self._tune_bufs = {}               # scratch operands of the autotuner are only needed while compiling
        if len(Plan._tune_measured) != n_tuned:
            Plan._tune_cache_io(save=True)

    # ------------------------------------------------------------------ helpers
    def _alloc(self, *shape, dtype=None, zero=False):
        f = torch.zeros if zero else torch.empty
        t = f(*shape, dtype=dtype or self.tdtype, device=self.device)
        self.keep.append(t)
        return t

    def _act_buf(self, a):
        if a.buf is None:
            a.buf = self._alloc(a.n, a.h, a.w, a.c, zero=a.c != a.c_valid)
        return a.buf

    def _act_grad(self, a):
        if a.grad is None:
            a.grad = self._alloc(a.n, a.h, a.w, a.c, zero=True)
        return a.grad

    def _pack(self, wt, n_out, n_in, strides, taps_rs, what):
        """Allocate a pack image and register the launch that (re)builds it from ``wt``."""
        nbytes = C.c_size_t(0)
        arr = _taps_array(taps_rs)
        check(self.lib.lh_pack_weight(None, None, C.byref(nbytes), n_out, n_in, *strides, len(taps_rs), arr, self.dt, None), what)
        buf = self._alloc(max(nbytes.value, 16), dtype=torch.uint8, zero=True)     # padding stays zero for ever
        if taps_rs and self._pack_regular(wt, buf, n_out, n_in, strides, taps_rs):
            return buf
        if taps_rs:
            it = _lib.PackItem()
            it.w, it.out, it.n_out, it.n_in, it.ntaps = wt.data_ptr(), buf.data_ptr(), n_out, n_in, len(taps_rs)
            it.so, it.si, it.sr, it.ss = strides
            for i, (r, q) in enumerate(taps_rs):
                it.r[i], it.s[i] = r, q
            self._pack_items.append(it)
            self.keep.append(wt)
        return buf

    # positions of lh_igemm's arguments inside a _Call.args tuple
    _IG = dict(desc=0, src=1, pack=2, dst=3, addend=4, addend_mask=5, bias=6, scale=7, shift=8, stats=9)

    def _pack_regular(self, wt, buf, n_out, n_in, strides, taps_rs):
        """Queue a pack of a plain [d0][d1][kH][kW] weight tensor for the LDS-tiled transposing pack kernel.
        Returns False when the tensor / strides are not of that form (the stem's staged image, oversize taps)."""
        if wt.dim() != 4 or not wt.is_contiguous() or len(taps_rs) > 16:
            return False
        d0, d1, r, s = wt.shape
        rs = r * s
        if 32 * (32 * rs + 2) * self.es > 64 * 1024:
            return False
        if tuple(strides) == (d1 * rs, rs, s, 1) and (n_out, n_in) == (d0, d1):
            row_is_d1 = 0
        elif tuple(strides) == (rs, d1 * rs, s, 1) and (n_out, n_in) == (d1, d0):
            row_is_d1 = 1
        else:
            return False
        conv = self._pack_convs.get(id(wt))
        if conv is None:
            conv = _lib.PackConv()
            conv.w, conv.d0, conv.d1, conv.rs, conv.npacks = wt.data_ptr(), d0, d1, rs, 0
            self._pack_convs[id(wt)] = conv
            self.keep.append(wt)
        if conv.npacks >= 5:
            return False
        o = conv.packs[conv.npacks]
        kstep = 128 // self.es
        o.out, o.row_is_d1, o.ntaps, o.kpad = buf.data_ptr(), row_is_d1, len(taps_rs), (n_in + kstep - 1) // kstep * kstep
        for i, (rr, ss) in enumerate(taps_rs):
            o.taps[i] = rr * s + ss
        conv.npacks += 1
        return True


    def _igemm(self, lst, d, src, pack, dst, addend, bias, stats, what, flops=0, produces=None, addend_mask=None):
        self.keep.append(d)
        c = _Call(self.lib.lh_igemm, (C.byref(d), _ptr(src), _ptr(pack), _ptr(dst), _ptr(addend), _ptr(addend_mask), _ptr(bias), 0, 0,
                                      _ptr(stats), self.dt), what)
        c.keep = d
        lst.append(c)
        if produces is not None:
            self._producers.setdefault(id(produces), []).append(c)
        return c

    def _dgrad(self, descs, dy, packs, x, what):
        """Data-gradient launches into x.grad (the first writer of a gradient buffer overwrites, later ones add)."""
        dx = self._act_grad(x)
        first = self._first_write(x)
        if not first and len(descs) > 1:          # accumulating: a phase without taps would add zeros -- drop it
            keep = [i for i, dd in enumerate(descs) if dd.ntaps > 0]
            descs, packs = [descs[i] for i in keep], [packs[i] for i in keep]
        addend, amask = (None if first else dx), None
        pend = self._masked_addend.pop(id(x), None)
        if pend is not None:
            # the identity-shortcut gradient of a residual tail, dout * mask, was NOT written by that tail's lh_fuse_bwd:
            # this first writer of dx adds it straight from dout (one tensor write and one read less per block)
            assert first
            addend, amask = pend
        kind = "masked" if amask is not None else ("plain" if addend is not None else None)
        if self._phase_rows(descs) > 0:
            self._tune(descs, addend=kind)
            self._igemm_phases(self.bwd, descs, dy, packs, dx, addend, None, None, what, addend_mask=amask)
            yield max(descs, key=lambda d: d.ntaps), sum(d.ntaps for d in descs)
            return
        for dd, pk in zip(descs, packs):
            # (networks with parallel branches: gated OUTSIDE their branch regions only -- HRNet's stem, layer1, transitions: 12.93 -> 12.84 ms.
            #  Inside them the members of a batch group run merged tiled launches whose reduce passes are merged too: carrying the gates into
            #  lh_igemm_multi was measured +0.26 ms SLOWER, round 6; a plan computes the same sums whether its branches run as groups or lanes)
            in_branch = self._emit_group > 1 or self._cur_lane != 0 or getattr(self, "_in_branch", False)
            gi = self._gate_info.get(id(x)) if (self.bn_gate and first and len(descs) == 1 and self.es == 2 and
                                                (self.n_lanes == 1 or (self.bn_gate_branches and not in_branch))) else None
            gkind, gbytes = self._gate_kind(gi, x, amask is not None), x.pixels * x.c * self.es
            self._tune([dd], addend=kind, role="dgrad", gate=(gkind, gbytes) if gkind else None)
            cfg = (C.c_int * 5)()
            if gkind and self.lib.lh_igemm_config(C.byref(dd), self.dt, cfg) == 0 and self._cfg_gateable(cfg, gkind, gbytes):
                # x = relu(BN(raw)) with this convolution as its only consumer -- or a residual tail relu(BN(raw) + r) whose other
                # consumer, the next tail's identity term, rides in as this launch's masked addend: the launch stores the ReLU-gated
                # gradient and the BatchNorm-backward partial sums of its tile (the node's backward skips its reduce pass)
                two = gkind == "mask2"
                rows = self.lib.lh_igemm_gated_rows(C.byref(dd), self.dt, 2 if two else 1)
                partial = self._alloc(rows * 2 * x.c, dtype=torch.float32)
                partial2 = self._alloc(rows * 2 * x.c, dtype=torch.float32) if two else None
                st = gi["st"]
                gate = _lib.BnBwdGate(gi["raw"].buf.data_ptr(), st["mean"].data_ptr(), st["invstd"].data_ptr(), st["scale"].data_ptr(),
                                      st["shift"].data_ptr(), partial.data_ptr(), _ptr(gi.get("mask")))
                if two:
                    st2 = gi["st2"]
                    gate.x2, gate.mean2, gate.invstd2, gate.partial2 = gi["raw2"].buf.data_ptr(), st2["mean"].data_ptr(), st2["invstd"].data_ptr(), partial2.data_ptr()
                self.keep += [dd, gate]
                c = _Call(self.lib.lh_igemm_gated, (C.byref(dd), _ptr(dy), _ptr(pk), _ptr(dx), _ptr(addend), _ptr(amask), C.byref(gate), self.dt),
                          what + " + BN-backward gate" + (" (mask bits)" if gi.get("mask") is not None else ""))
                c.keep = dd
                c.ig = dict(src=1, dst=3, addend=4, addend_mask=5)
                self.bwd.append(c)
                self._gated[id(x)] = (partial, rows, partial2)
            else:
                self._igemm(self.bwd, dd, dy, pk, dx, addend, None, None, what, addend_mask=amask)
            yield dd, dd.ntaps

    def _gate_meta(self, dd, x):
        """(kernel name, extra algorithmic bytes) of the data gradient just emitted when it is a gated launch (lh_igemm_gated): the persistent
        kernels have gate instantiations of their own (igemm_pw_kernel<.., true, terms>, conv3x3_direct_kernel<.., true, true>), and the
        epilogue reads the BatchNorm input of every gated term (what the reduce pass of lh_fuse_bwd no longer reads) plus the mask bits."""
        c = self.bwd[-1]
        if getattr(c, "fn", None) is not self.lib.lh_igemm_gated:
            return None, 0.0
        g = self._gated[id(x)]
        terms = 2 if len(g) > 2 and g[2] is not None else 1
        name = self._kname(dd, stats=True)
        if name.startswith("igemm_pw_kernel"):
            name = name[:-1] + f", {terms}>"
        elif name.startswith("conv3x3_direct_kernel"):
            name = name[:-1] + ", true>"
        return name, float(terms) * x.pixels * x.c * self.es + (x.pixels * x.c / 8 if "mask" in c.what else 0.0)

    def _gate_kind(self, gi, x, masked_addend):
        """May the first writer of x.grad, a data gradient, take the BatchNorm-backward gate of x's node (lh_igemm_gated)?  None | 'x' | 'mask'.
        'x': a single-term node a = relu(BN(raw)) whose only consumer is this convolution.  'mask' (round 6): a residual tail
        relu(BN(raw) + r) -- this convolution is its only consumer, or the other one is the next tail, whose identity gradient has been
        folded into this launch as its masked addend."""
        if gi is None or x.c != x.c_valid:
            return None
        uses = len(self._uses.get(id(x), []))
        if gi.get("mask") is not None:
            ok = self.bn_gate_tail and (uses == 1 or (uses == 2 and masked_addend))
            return None if not ok else "mask2" if gi.get("raw2") is not None else "mask"
        return "x" if uses == 1 else None

    def _cfg_gateable(self, cfg, kind, nbytes):
        """Does kernel configuration cfg take the gate for a tensor of nbytes?  The tiled kernels up to LH_BN_GATE_MAX_MB (measured,
        rounds 4-6: beyond it the epilogue's read of raw costs a tile-per-workgroup launch more than the reduce pass it replaces), the
        persistent kernels (pointwise, direct 3x3: streams of independent waves, the extra read rides with the others) up to
        LH_BN_GATE_PW_MAX_MB; tails up to LH_BN_GATE_TAIL_MAX_MB on either."""
        pw, tiled = cfg[2] in (1, 100), (2 <= cfg[2] < 10 or 20 <= cfg[2] < 40)
        if not (tiled or (pw and self.bn_gate_pw)):
            return False
        if kind == "mask2":                # two BatchNorm terms (a projection shortcut): the pointwise kernel only
            return cfg[2] == 1 and self.bn_gate_tail2 and nbytes <= self.bn_gate_tail_bytes
        if kind == "mask":
            return nbytes <= (self.bn_gate_tail_bytes if pw else min(self.bn_gate_tail_bytes, self.bn_gate_tiled_tail_bytes))
        return nbytes <= (self.bn_gate_pw_bytes if pw else self.bn_gate_bytes)

    def _patch(self, call, relu=None, **ptrs):
        a = list(call.args)
        ig = call.ig or self._IG
        for k, v in ptrs.items():
            a[ig[k]] = v
        call.args = tuple(a)
        if relu is not None:
            for d in (call.keep if isinstance(call.keep, list) else [call.keep]):
                d.relu = int(relu)

    _IGP = dict(src=2, dst=4, addend=5, addend_mask=6, bias=7, scale=8, shift=9, stats=10)      # lh_igemm_phases argument positions

    def _phase_rows(self, descs):
        """Rows of the stats slab ONE phase of a batched launch writes, or -1 when the phases cannot be batched."""
        if len(descs) < 2 or len(descs) > 4:
            return -1
        arr = (C.POINTER(IgemmDesc) * len(descs))(*[C.pointer(d) for d in descs])
        return self.lib.lh_igemm_phases_rows(arr, len(descs), self.dt)

    def _igemm_phases(self, lst, descs, src, packs, dst, addend, bias, stats, what, produces=None, addend_mask=None):
        """The sub-pixel phases of a stride-2 transposed form as one launch (lh_igemm_phases)."""
        arr = (C.POINTER(IgemmDesc) * len(descs))(*[C.pointer(d) for d in descs])
        parr = (C.c_void_p * len(descs))(*[pk.data_ptr() if pk is not None else None for pk in packs])
        self.keep += [arr, parr] + list(descs)
        c = _Call(self.lib.lh_igemm_phases, (arr, len(descs), _ptr(src), parr, _ptr(dst), _ptr(addend), _ptr(addend_mask), _ptr(bias), 0, 0,
                                             _ptr(stats), self.dt), what + f" ({len(descs)} phases)")
        c.keep, c.ig = list(descs), self._IGP
        lst.append(c)
        if produces is not None:
            self._producers.setdefault(id(produces), []).append(c)
        return c

    def _copy4(self, dst, src, what, lane=0):
        """dst.copy_(src) for two fp32 views of equal shape (<= 4 dims, any strides) as one C-ABI launch."""
        assert dst.dtype == src.dtype == torch.float32 and tuple(dst.shape) == tuple(src.shape) and dst.dim() <= 4
        pad = 4 - dst.dim()
        shape = (C.c_int * 4)(*([1] * pad + list(dst.shape)))
        ds = (C.c_long * 4)(*([0] * pad + list(dst.stride())))
        ss = (C.c_long * 4)(*([0] * pad + list(src.stride())))
        return _Call(self.lib.lh_copy_strided_f32, (dst.data_ptr(), src.data_ptr(), shape, ds, ss), what, keep=(shape, ds, ss, dst, src), lane=lane)

    def _bias_grad(self, dy, y, cout, gb_):
        """d(bias) of a convolution / transposed convolution inside the network = the per-channel sum of its output
        gradient (NHWC, run precision): lh_channel_sum_nhwc, fp64 partial sums in a fixed order."""
        ws = self._alloc(self.lib.lh_channel_sum_workspace_bytes(cout), dtype=torch.uint8)
        return _Call(self.lib.lh_channel_sum_nhwc, (dy.data_ptr(), y.pixels, cout, y.c, gb_.data_ptr(), ws.data_ptr(), self.dt), "bias grad", lane=1)

    def _kname(self, d, wgrad=None, stats=False):
        """Kernel instantiation name as rocprofv3 prints it (for roofline attribution)."""
        t = {"fp32": "float", "bf16": "__bf16", "fp16": "_Float16"}[self.precision]
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        if wgrad is None:
            cfg = (C.c_int * 5)()
            check(self.lib.lh_igemm_config(C.byref(d), self.dt, cfg), "lh_igemm_config")
            bm, bp, depth, kb = cfg[0], cfg[1], cfg[2], cfg[3]
            if depth == 1:
                return f"igemm_pw_kernel<{t}, {bm}, {kb}, {bp // 16}, {'true' if stats else 'false'}>"
            if depth == 100:
                return f"conv3x3_direct_kernel<{t}, {kb}, {'true' if stats else 'false'}>"
            wc, wp = {(256, 256): (2, 4), (128, 256): (2, 2), (256, 128): (4, 2), (128, 128): (2, 2), (128, 64): (4, 1), (64, 128): (1, 4), (64, 64): (2, 2)}[(bm, bp)]
            if 30 <= depth < 40:                # the K-split wave-pair forms (igemm_ring_cfgs.h)
                return f"igemm_ring_ksplit_kernel<{t}, {bm}, {bp}, {wc}, {wp}, {depth - 30}, {kb}>"
            if 20 <= depth < 30:                # the dense-wave forms (eight waves on the 4-wave tiles, igemm_ring_cfgs.h)
                wc, wp = {(128, 128): (2, 4), (128, 64): (4, 2), (64, 128): (2, 4), (64, 64): (2, 4), (128, 256): (2, 4), (256, 128): (4, 2)}[(bm, bp)]
                depth -= 20
            if depth:
                return f"igemm_ring_kernel<{t}, {bm}, {bp}, {wc}, {wp}, {depth}, {kb}>"
            return f"igemm_kernel<{t}, {bm}, {bp}, {wc}, {wp}>"
        r = C.c_int(0)
        check(self.lib.lh_wgrad_tile(C.byref(d), wgrad[0], wgrad[1], self.dt, C.byref(a), C.byref(b), C.byref(c), C.byref(r)), "lh_wgrad_tile")
        wo, wi = {(256, 256): (2, 4), (128, 128): (2, 2), (128, 64): (4, 1), (64, 128): (1, 4), (64, 64): (2, 2)}[(a.value, b.value)]
        if r.value:
            return f"wgrad_ring_kernel<{t}, {a.value}, {b.value}, {wo}, {wi}, {r.value % 10}, {r.value // 10}>"
        return f"wgrad_kernel<{t}, {a.value}, {b.value}, {wo}, {wi}>"


    def _first_write(self, a):
        """True the first time a gradient buffer is produced in the backward list (every writer calls this once)."""
        n = self._nwrites.get(id(a), 0)
        self._nwrites[id(a)] = n + 1
        return n == 0

    def _stats_for(self, y, descs):
        rows = [self.lib.lh_igemm_stats_rows(C.byref(d), self.dt) for d in descs]
        total = sum(rows)
        nbytes = self.lib.lh_bn_stats_slab_bytes(total, y.c)
        y.stats = self._alloc((nbytes + 3) // 4, dtype=torch.float32)
        y.stats_rows = total
        offs, o = [], 0
        for r in rows:
            offs.append(o)
            o += r * 2 * y.c * 4
        return offs

    # ------------------------------------------------------------------ compile
    def _compile(self):
        consumers_bn = set()
        for kind, nd in self.nodes:
            if kind == "fuse":
                for a, bn, _ in nd["terms"]:
                    if bn is not None:
                        consumers_bn.add(id(a))
        self._bn_inputs = consumers_bn
        self._nwrites = {}
        self._masked_addend = {}           # id(activation) -> (dout, relu mask bits) a later data-gradient launch adds
        # BatchNorm-backward gate (lh_igemm_gated): id(activation a = relu(BN(x))) -> what the data gradient that writes a.grad
        # needs (recorded by _c_fuse), and id(a) -> (partial sums, rows) once such a launch has been planned (read by the node's
        # backward, which then skips its reduce pass).  LH_BN_GATE=0: off.
        self._gate_info, self._gated = {}, {}
        self._bnrelu_info = {}             # id(a = relu(BN(x))) -> its lh_fuse_fwd call and BN state (training plans; _c_maxpool)
        self.bn_gate = os.environ.get("LH_BN_GATE", "1") != "0"
        self.bn_gate_bytes = float(os.environ.get("LH_BN_GATE_MAX_MB", "9")) * (1 << 20)
        self.bn_gate_pw = os.environ.get("LH_BN_GATE_PW", "1") != "0"               # round 6: the pointwise kernel's epilogue takes the gate too
        self.bn_gate_pw_bytes = float(os.environ.get("LH_BN_GATE_PW_MAX_MB", "1024")) * (1 << 20)
        self.bn_gate_tail = os.environ.get("LH_BN_GATE_TAIL", "1") != "0"           # round 6: residual tails (sign from the stored mask bits)
        self.bn_gate_tail_bytes = float(os.environ.get("LH_BN_GATE_TAIL_MAX_MB", "1024")) * (1 << 20)
        self.bn_gate_tiled_tail_bytes = float(os.environ.get("LH_BN_GATE_TILED_TAIL_MAX_MB", "1024")) * (1 << 20)
        self.bn_gate_tail2 = os.environ.get("LH_BN_GATE_TAIL2", "1") != "0"         # ... tails with a projection shortcut (two BatchNorm terms)
        self.bn_gate_branches = os.environ.get("LH_BN_GATE_BRANCHES", "1") != "0"   # ... in networks with parallel branches (HRNet), outside the branch regions
        # consumers of every activation in forward order: (kind, node) -- backward visits them in reverse
        self._uses = {}
        for kind, nd in self.nodes:
            if kind in ("conv", "deconv", "maxpool"):
                self._uses.setdefault(id(nd["x"]), []).append((kind, nd))
            elif kind == "fuse":
                for a, _, _ in nd["terms"]:
                    self._uses.setdefault(id(a), []).append((kind, nd))
            elif kind == "output":
                self._uses.setdefault(id(nd["y"]), []).append((kind, nd))
        items = self._batch_order()        # lists of node indices; more than one = a batch group
        bwd_blocks, group_forced = {}, {}
        for item in items:
            gid = None
            if len(item) > 1:
                self._n_groups += 1
                gid = self._n_groups
                self._forced = self._tune_group([self.nodes[i][1] for i in item]) if self.nodes[item[0]][0] == "conv" else None
                group_forced[item[0]] = self._forced
            for j, i in enumerate(item):
                (kind, nd), lane = self.nodes[i], self.node_lanes[i]
                if self._forced is not None:
                    self._forced["member"] = j
                blk = []
                n0 = len(self.fwd)
                getattr(self, "_c_" + kind)(nd, blk)
                for k, c in enumerate(self.fwd[n0:]):
                    c.slane = lane
                    if gid is not None and isinstance(c, _Call):
                        c.mtag = (gid, "f", j, k)
                out_act = nd.get("y", nd.get("out")) if isinstance(nd, dict) else nd
                if out_act is not None:
                    self._ready[id(out_act)] = len(self.fwd)      # list position from which this activation is complete
                bwd_blocks[i] = blk
            self._forced = None
        # backward list: node blocks in reverse order; accumulate flags resolved in that order
        if self._pack_items:       # every weight pack of the model is rebuilt by ONE launch
            arr = (_lib.PackItem * len(self._pack_items))(*self._pack_items)
            table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
            chunk, kstep = self.lib.lh_pack_chunk_elems(), 128 // self.es
            c_item, c_start = [], []
            for i, it in enumerate(self._pack_items):
                total = (it.n_out + 127) // 128 * 128 * it.ntaps * ((it.n_in + kstep - 1) // kstep * kstep)
                for s0 in range(0, total, chunk):
                    c_item.append(i)
                    c_start.append(s0)
            t_item = torch.tensor(c_item, dtype=torch.int32, device=self.device)
            t_start = torch.tensor(c_start, dtype=torch.int64, device=self.device)
            self.keep += [table, t_item, t_start]
            self.packs.append(_Call(self.lib.lh_pack_weights_multi,
                                    (table.data_ptr(), t_item.data_ptr(), t_start.data_ptr(), len(c_item), self.dt), "weight packs"))
        self._late_packs = None    # (pack buffer pointers of the late group, index of the conv whose first use is the fork point)
        if self._pack_convs:       # regular conv / deconv weights: the tiled transposing pack kernel
            convs = list(self._pack_convs.values())        # in the order the forward pass first uses them
            # Training plans split the launch: the layers the forward pass reaches LATE and that hold most of the bytes
            # (R50: stage 4 + the head's transposed convolutions, 75 % of the parameters) are packed by a second launch that
            # runs under the latency-bound middle of the forward pass instead of beside the HBM-bound stem and stage 1.
            groups = [convs]
            if self.with_bwd and len(convs) >= 16 and os.environ.get("LH_LATE_PACK", "1") != "0":
                size = [cv.d0 * cv.d1 * cv.rs for cv in convs]
                total, acc, cut = sum(size), 0, len(convs)
                while cut > 0 and acc + size[cut - 1] <= 0.8 * total:
                    cut -= 1
                    acc += size[cut]
                if 8 <= cut < len(convs) and acc >= 0.5 * total:
                    groups = [convs[:cut], convs[cut:]]
                    fork_conv = max(1, cut - max(8, int(0.35 * len(convs))))
                    self._late_packs = ({convs[i].packs[k].out for i in range(cut, len(convs)) for k in range(convs[i].npacks)},
                                        {convs[fork_conv].packs[k].out for k in range(convs[fork_conv].npacks)})
            for gi, grp in enumerate(groups):
                arr = (_lib.PackConv * len(grp))(*grp)
                table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
                c_conv, c_t0, c_t1 = [], [], []
                for i, cv in enumerate(grp):
                    for a in range((cv.d0 + 31) // 32):
                        for b in range((cv.d1 + 31) // 32):
                            c_conv.append(i); c_t0.append(a); c_t1.append(b)
                tabs = [torch.tensor(v, dtype=torch.int32, device=self.device) for v in (c_conv, c_t0, c_t1)]
                self.keep += [table] + tabs
                call = _Call(self.lib.lh_pack_weights_tiled,
                             (table.data_ptr(), tabs[0].data_ptr(), tabs[1].data_ptr(), tabs[2].data_ptr(), len(c_conv),
                              max(cv.rs for cv in grp), self.dt), "weight packs (tiled)" + (", late group" if gi else ""))
                call.lane = gi           # 1 = the late group (refresh_packs(overlap=True) defers it to the 'packfork2' marker)
                self.packs.append(call)
        self.bwd_marks = []        # (end index in self.bwd, parameter names whose gradient is final there)
        if self.with_bwd:
            gcount = self._n_groups
            for item in reversed(items):
                gid = None
                if len(item) > 1:
                    gcount += 1
                    gid = gcount
                flush_lanes = []
                self._emit_group = len(item)
                self._forced = group_forced.get(item[0])          # the data gradients are tuned while the blocks are emitted
                for j, i in enumerate(item):
                    (kind, nd), blk, lane = self.nodes[i], bwd_blocks[i], self.node_lanes[i]
                    self._cur_lane, self._in_branch = lane, self.node_branch[i]
                    if self._forced is not None:
                        self._forced["member"] = j
                    if kind == "fork" and self.wgrad_group > 0:      # end of a branch region in backward order: every lane hands
                        for src in sorted(self._pend):                # its group over before the main stream joins
                            self._flush_wgrads(src)
                    n0 = len(self.bwd)
                    w0 = len(self._pending()["calls"]) if self.wgrad_group > 0 else 0
                    for emit in blk:
                        emit()
                    for k, c in enumerate(self.bwd[n0:]):
                        c.slane = lane
                        if gid is not None and isinstance(c, _Call):
                            c.mtag = (gid, "b", j, k)
                    if gid is not None and self.wgrad_group > 0:
                        for k, c in enumerate(self._pending()["calls"][w0:]):
                            if isinstance(c, _Call):
                                c.mtag = (gid, "w", j, k)
                    names = []
                    if kind in ("conv", "deconv"):
                        wnames = [nd["w"] + ".weight"] + ([nd["bias"]] if nd["bias"] else [])
                        if self.wgrad_group > 0:
                            p = self._pending()
                            p["names"] += wnames
                            p["layers"] += 1
                            p["bytes"] += sum(self.params[k].numel() * 4 for k in wnames)
                            # same rule as parallel.wgrad_group_cuts (tested on the CPU with the real parameter sizes)
                            if p["layers"] >= self.wgrad_group or (self.wgrad_bucket_bytes and p["bytes"] >= self.wgrad_bucket_bytes):
                                flush_lanes.append(lane)          # after the whole item: a batch group hands over together
                        else:
                            names += wnames
                    elif kind == "fuse":
                        for _, bn, _ in nd["terms"]:
                            if bn is not None:
                                names += [bn + ".weight", bn + ".bias"]
                    if names:
                        self.bwd_marks.append((len(self.bwd), names))
                self._forced = None
                for lane in dict.fromkeys(flush_lanes):
                    self._cur_lane = lane
                    self._flush_wgrads(lane)
            for src in sorted(self._pend):
                self._flush_wgrads(src, spread=os.environ.get("LH_TAIL_SPREAD", "1") != "0")
            self.bwd_marks.sort(key=lambda m: m[0])
            # two workspaces: the weight-gradient chain may run concurrently with the BN-backward chain
            # (stream lanes run concurrently: each lane has its own pair)
            lanes = list(range(self.n_lanes if self.use_lanes else 1)) + ([-1 - i for i in range(self._w_lanes)] if self.wgrad_group > 0 else [])
            need_w = {L: max([nb for _, nb, l in self._ws_users if (l if self.use_lanes else 0) == L] + [256]) for L in lanes}
            ws_w = {L: self._alloc(need_w[L], dtype=torch.uint8) for L in lanes}      # ONE shared, cache-resident slab per stream
            ws_f = {L: self._alloc(max(self._ws_fuse, 256), dtype=torch.uint8) for L in lanes}
            for setter, nbytes, lane in self._ws_users:
                L = lane if self.use_lanes else 0
                setter(ws_w[L].data_ptr())
            for setter, lane in self._ws_users_fuse:
                setter(ws_f[lane if self.use_lanes else 0].data_ptr())
        self._attach_l2_touch()
        if self._n_groups:
            self._merge_groups()
        # where the forward list first reads a pack written by the tiled pack launch (refresh_packs(overlap=True))
        self._pack_event, self._packjoin_at, self._pack_stream = None, None, None
        if self.with_bwd and any(getattr(c, "fn", None) is self.lib.lh_pack_weights_tiled for c in self.packs):
            convs = (self.lib.lh_igemm, self.lib.lh_igemm_multi, self.lib.lh_igemm_phases, self.lib.lh_igemm_phases_head)
            for i, c in enumerate(self.fwd):
                if isinstance(c, _Call) and any(c.fn is f for f in convs) and not c.what.endswith("stem fwd"):
                    self.fwd.insert(i, _Marker("packjoin"))
                    self._packjoin_at = i
                    self._pack_stream = torch.cuda.Stream(device=self.device)
                    break
            if self._late_packs is not None and self._packjoin_at is not None:
                late, fork_at = self._late_packs
                first = lambda ptrs: next((i for i, c in enumerate(self.fwd) if isinstance(c, _Call) and self._call_packs(c) & ptrs), None)
                j, f = first(late), first(fork_at)
                if j is not None and f is not None and self._packjoin_at < f < j:
                    self.fwd.insert(j, _Marker("packjoin2"))
                    self.fwd.insert(f, _Marker("packfork2"))
                else:
                    self._late_packs = None
        self._pack_late, self._pack_event2 = None, None

    def _attach_l2_touch(self):
        """Training plans: an elementwise BatchNorm / ReLU pass (lh_fuse_fwd) that is followed on its stream by a tiled
        convolution warms that convolution's weight pack in L2 at its tail (lh_fuse_desc.l2_touch; bn.hip lh_l2_touch).  Every
        workgroup of such a convolution walks the same weight slab stage by stage, at once: each stage waits for lines no XCD has
        seen yet (profiles/r05_ingest_ladder.txt, sitting 6: the complete K loop of the stage-3 3x3 takes 21.3 us, 19.1 us with the
        pack in L2).  Only where the pack fits beside the pass's own stream in the 4 MB L2 of an XCD (LH_L2_TOUCH_MAX_MB, default 3;
        LH_L2_TOUCH=0: off).  Members of HRNet's batch groups only with LH_L2_TOUCH_GROUPS=1 (measured slightly slower)."""
        if os.environ.get("LH_L2_TOUCH", "1") == "0" or not self.training:
            return
        lim = float(os.environ.get("LH_L2_TOUCH_MAX_MB", "3")) * (1 << 20)
        touch_all = os.environ.get("LH_L2_TOUCH", "1") == "2"      # experiment: the persistent kernels' panels too
        lib, ig, n = self.lib, self._IG, 0
        # members of HRNet's batch groups too (their descriptors travel into the merged calls)?  MEASURED (HRNet-W32 bs 32 fp16): 13.04-13.06
        # ms with them, 13.01-13.02 without, 13.05-13.07 with no touch at all: off by default
        grouped = os.environ.get("LH_L2_TOUCH_GROUPS", "0") == "1"

        def attach(lst, fuse_fn, look):
            nonlocal n
            for i, c in enumerate(lst):
                if not isinstance(c, _Call) or c.fn is not fuse_fn or c.lane or (c.mtag is not None and not grouped):
                    continue
                # the next convolution on this call's stream lane: right behind it (single launches), or the same member of the next
                # position of a batch group (the other members' launches sit in between until _merge_groups merges them)
                nxt = None
                for d in lst[i + 1:i + 1 + (look if c.mtag is None else 4 * look)]:
                    if not isinstance(d, _Call) or d.lane or d.fn is lib.lh_bn_finalize:
                        continue
                    if c.mtag is None or d.slane == c.slane:
                        nxt = d
                        break
                if nxt is None or nxt.fn is not lib.lh_igemm or nxt.slane != c.slane or (nxt.mtag is None) != (c.mtag is None):
                    continue
                d = nxt.keep
                if (d.cfg[2] in (1, 100) and not touch_all) or not nxt.args[ig["pack"]]:    # pointwise / direct kernels fetch their panel once per workgroup
                    continue
                kstep = 128 // self.es
                nbytes = (d.cout + 127) // 128 * 128 * d.ntaps * ((d.k_run + kstep - 1) // kstep * kstep) * self.es
                if not 0 < nbytes <= lim:
                    continue
                fd = c.args[0]._obj
                fd.l2_touch, fd.l2_touch_bytes = nxt.args[ig["pack"]], nbytes
                n += 1
        attach(self.fwd, lib.lh_fuse_fwd, 3)
        # the same in the backward list: the BatchNorm / ReLU backward of a node (lh_fuse_bwd: its last apply pass) in front of the
        # data gradient that consumes the gradient it wrote
        attach(self.bwd, lib.lh_fuse_bwd, 2)
        self._n_l2_touch = n

    def _call_packs(self, c):
        """Pack buffers a forward convolution call reads (addresses)."""
        lib = self.lib
        if c.fn is lib.lh_igemm:
            return {c.args[2]}
        if c.fn is lib.lh_igemm_multi:
            return {c.args[0][i].wpack for i in range(c.args[1])}
        if c.fn is lib.lh_igemm_phases or c.fn is lib.lh_igemm_phases_head:
            return {c.args[3][i] for i in range(c.args[1])}
        return set()

    def _c_nop(self, nd, blk):
        pass


    def _c_fork(self, nd, blk):
        self.fwd.append(_Marker("fork"))
        self._regions = getattr(self, "_regions", [])
        self._regions.append([self.fwd[-1], None])
        blk.append(lambda: self.bwd.append(_Marker("join")))          # backward walks the region in reverse

    def _c_join(self, nd, blk):
        self.fwd.append(_Marker("join"))
        self._regions[-1][1] = self.fwd[-1]
        blk.append(lambda: self.bwd.append(_Marker("fork")))

    def _in_closed_region(self, call):
        """True when `call` sits inside a fork..join region that is already closed: work moved into it from a later node
        (eval-mode folding of a cross-branch sum) would read another lane's output unordered."""
        if not self.use_lanes:
            return False
        i = self.fwd.index(call)
        return any(r[1] is not None and self.fwd.index(r[0]) < i < self.fwd.index(r[1]) for r in getattr(self, "_regions", []))

    def _c_input(self, a, blk):
        pass        # the consumer (stem conv) owns the image transform

    def _c_input_act(self, a, blk):
        self.in_act = a
        self._act_buf(a)
        if self.with_bwd:
            self._act_grad(a)

    def _c_output(self, nd, blk):
        y = nd["y"]
        if getattr(self, "_head_fused", False):           # the fused head wrote the fp32 NCHW heat-map itself (_fuse_head)
            return
        self.out_nchw = self._alloc(y.n, y.c_valid, y.h, y.w, dtype=torch.float32)
        self.fwd.append(_Call(self.lib.lh_nhwc_to_nchw_f32, (y.buf.data_ptr(), self.out_nchw.data_ptr(), y.n, y.h, y.w, y.c_valid, y.c, self.dt), "output transform"))
        if self.with_bwd:
            self.dout_nchw = self._alloc(y.n, y.c_valid, y.h, y.w, dtype=torch.float32, zero=True)

            def emit():
                g = self._act_grad(y)
                self._first_write(y)
                self.bwd.append(_Call(self.lib.lh_nchw_f32_to_nhwc, (self.dout_nchw.data_ptr(), g.data_ptr(), y.n, y.h, y.w, y.c_valid, y.c, self.dt), "dheat transform"))
            blk.append(emit)


    # ---- convolution ---------------------------------------------------------------------------
    def _c_conv(self, nd, blk):
        x, y, k, s, p = nd["x"], nd["y"], nd["k"], nd["s"], nd["p"]
        wt = self.params[nd["w"] + ".weight"]
        cout, cin = wt.shape[0], wt.shape[1]
        ybuf = self._act_buf(y)
        bias = None
        if nd["bias"]:
            bias = self._alloc(y.c, dtype=torch.float32, zero=True)
            bsrc = self.params[nd["bias"]]
            self.packs.append(self._copy4(bias[:cout], bsrc.detach(), "bias pad"))
        all_rs = [(r, q) for r in range(k) for q in range(k)]
        if x.is_image:
            self._c_stem(nd, blk, bias)
            return
        xbuf = self._act_buf(x)
        taps = [(r - p, q - p) for r, q in all_rs]
        d = _desc(x.n, x.h, x.w, x.c, cin, y.h, y.w, s, s, y.c, y.h, y.w, 1, 1, 0, 0, y.c, taps)
        pack = self._pack(wt, cout, cin, (cin * k * k, k * k, k, 1), all_rs, nd["w"] + " fwd pack")
        if self._fuse_head(nd, pack, bias):
            return
        stats_ptr = None
        self._tune([d], with_stats=id(y) in self._bn_inputs and self.training, role="fwd")
        if id(y) in self._bn_inputs and self.training:
            self._stats_for(y, [d])
            stats_ptr = y.stats
        flops = 2.0 * y.pixels * cout * cin * k * k
        self._igemm(self.fwd, d, xbuf, pack, ybuf, None, bias, stats_ptr, nd["w"] + " fwd", produces=y)
        self.profile_meta.append(("fwd", self.fwd[-1], self._kname(d, stats=stats_ptr is not None), flops, (x.pixels * x.c + y.pixels * y.c) * self.es))
        if not self.with_bwd:
            return
        # --- backward: weight gradient, then data gradient
        rs_arr = _taps_array(all_rs)
        gw = self.grads[nd["w"] + ".weight"]
        pad_out = y.c != cout
        gtmp = self._alloc(y.c, cin, k, k, dtype=torch.float32) if pad_out else gw

        def tune_launch(xp, dyp, ws, grad, sp):
            check(self.lib.lh_wgrad_fused(C.byref(d), 0, xp, dyp, y.c, y.c, cin, ws, grad, cin * k * k, k * k, k, 1, rs_arr, 0, self.dt, sp),
                  "autotune lh_wgrad_fused")
        self._tune_wgrad(d, y.c, cin, y.c, tune_launch, y.c * cin * k * k)
        slab_bytes = self.lib.lh_wgrad_workspace_bytes(C.byref(d), y.c, cin, self.dt)
        self._ws_wgrad = max(self._ws_wgrad, slab_bytes)
        dpacks, ddescs = [], []
        if x.needs_grad:
            for ph in range(s):
                for pw in range(s):
                    sub = [(r, q) for r, q in all_rs if (ph + p - r) % s == 0 and (pw + p - q) % s == 0]
                    tp = [((ph + p - r) // s, (pw + p - q) // s) for r, q in sub]
                    gh, gw_ = (x.h - ph + s - 1) // s, (x.w - pw + s - 1) // s
                    dd = _desc(y.n, y.h, y.w, y.c, y.c if pad_out else cout, gh, gw_, 1, 1, x.c, x.h, x.w, s, s, ph, pw, x.c, tp)
                    ddescs.append(dd)
                    dpacks.append(self._pack(wt, cin, cout, (k * k, cin * k * k, k, 1), sub, nd["w"] + " dgrad pack"))

        def emit():
            dy = self._act_grad(y)
            # weight gradient + fold of the pixel splits as ONE C-ABI call (wgrad kernel, then reduce kernel)
            a = [C.byref(d), 0, xbuf.data_ptr(), dy.data_ptr(), y.c, y.c, cin, 0, gtmp.data_ptr(), cin * k * k, k * k, k, 1, rs_arr, 0, self.dt]
            tail = 1
            cw = _Call(self.lib.lh_wgrad_fused, None, nd["w"] + " wgrad", keep=rs_arr, lane=1)
            cw.keep_desc = d
            cw.wargs, cw.wbufs = a, (xbuf, dy)

            def set_ws(ptr, cw=cw, a=a):
                a[7] = ptr
                cw.args = tuple(a)
            cw.ws_ent = self._ws_note(set_ws, slab_bytes)
            wl = self._wl()
            wl.append(cw)
            self.profile_meta.append(("bwd", wl[-1], self._kname(d, (y.c, cin)), flops, (x.pixels * x.c + y.pixels * y.c) * self.es))
            if pad_out:
                wl.append(self._copy4(gw, gtmp[:cout].view_as(gw), "head grad crop", lane=tail))
            if nd["bias"]:
                gb_ = self.grads[nd["bias"]]
                if y is self.out_act:      # the head: reduce the contiguous fp32 NCHW gradient instead of strided bf16
                    dn = self.dout_nchw
                    ws = self._alloc(self.lib.lh_channel_sum_workspace_bytes(dn.shape[1]), dtype=torch.uint8)
                    wl.append(_Call(self.lib.lh_channel_sum_nchw, (dn.data_ptr(), dn.shape[0], dn.shape[1], dn.shape[2] * dn.shape[3],
                                                                          gb_.data_ptr(), ws.data_ptr()), "head bias grad"))
                else:
                    wl.append(self._bias_grad(dy, y, cout, gb_))
            if x.needs_grad:
                for dd, ntaps in self._dgrad(ddescs, dy, dpacks, x, nd["w"] + " dgrad"):
                    batched = ntaps != dd.ntaps or (len(ddescs) > 1 and self.bwd[-1].ig is not None)
                    gname, gbytes = self._gate_meta(dd, x)
                    self.profile_meta.append(("bwd", self.bwd[-1], gname or self._kname(dd), 2.0 * dd.n * dd.ho * dd.wo * cin * cout * ntaps,
                                              gbytes + ((x.pixels * x.c + y.pixels * y.c) * self.es if batched else
                                                        (dd.n * dd.ho * dd.wo * x.c + y.pixels * y.c / (s * s)) * self.es)))
        blk.append(emit)

    def _c_stem(self, nd, blk, bias):
        """C_in = 3 convolution: the image is stored as zero-padded NHWC4 and every kernel ROW is
        one tap whose K run covers the k pixels x 4 channels that are contiguous in memory."""
        x, y, k, s, p = nd["x"], nd["y"], nd["k"], nd["s"], nd["p"]
        wt = self.params[nd["w"] + ".weight"]
        cout = wt.shape[0]
        kr = (k * 4 + 7) // 8 * 8
        hp, wp = x.h + 2 * p, x.w + 2 * p + 2
        assert (y.w - 1) * s * 4 + kr <= wp * 4
        self.img_nchw = self._alloc(x.n, 3, x.h, x.w, dtype=torch.float32)
        img = self._alloc(x.n, hp, wp, 4)
        self.img_nhwc4, self.img_pad, self.img_wp = img, p, wp
        self.fwd.append(_Call(self.lib.lh_image_to_nhwc4, (self.img_nchw.data_ptr(), img.data_ptr(), x.n, x.h, x.w, p, wp, self.dt), "image transform"))
        self._image_call_index = len(self.fwd) - 1
        stage = self._alloc(cout, k, kr // 4, 4, dtype=torch.float32, zero=True)
        self.packs.append(self._copy4(stage[:, :, :k, :3], wt.detach().permute(0, 2, 3, 1), "stem weight staging"))
        rows = [(r, 0) for r in range(k)]
        pack = self._pack(stage, cout, kr, (k * kr, 1, kr, 0), rows, nd["w"] + " stem pack")
        d = _desc(x.n, hp, wp, 4, kr, y.h, y.w, s, s, y.c, y.h, y.w, 1, 1, 0, 0, y.c, rows)
        ybuf = self._act_buf(y)
        stats_ptr = None
        flops = 2.0 * y.pixels * cout * 3 * k * k
        direct = (self.training and id(y) in self._bn_inputs and self.es == 2 and (k, s, p, cout, y.c) == (7, 2, 3, 64, 64) and bias is None
                  and os.environ.get("LH_STEM_DIRECT", "1") != "0")
        if direct:
            # training stem (pose_resnet.py:151-152) on the direct kernel: weights in registers, a tile's input patch in LDS,
            # raw convolution output + one statistics row per workgroup (stem_pool.hip, lh_stem_conv)
            rows_ = self.lib.lh_stem_conv_rows(x.n, y.h, y.w)
            y.stats = self._alloc((self.lib.lh_bn_stats_slab_bytes(rows_, y.c) + 3) // 4, dtype=torch.float32)
            y.stats_rows = rows_
            self.keep.append(d)
            c_ = _Call(self.lib.lh_stem_conv, (img.data_ptr(), x.n, hp, wp, pack.data_ptr(), ybuf.data_ptr(), y.stats.data_ptr(), y.h, y.w, self.dt),
                       nd["w"] + " stem fwd (direct)")
            self.fwd.append(c_)
            self._producers.setdefault(id(y), []).append(c_)
            self.profile_meta.append(("fwd", self.fwd[-1], "stem_conv_kernel", flops, (x.pixels * 4 + y.pixels * y.c) * self.es))
        else:
            self._tune([d], with_stats=id(y) in self._bn_inputs and self.training)
            if id(y) in self._bn_inputs and self.training:
                self._stats_for(y, [d])
                stats_ptr = y.stats
            self._igemm(self.fwd, d, img, pack, ybuf, None, bias, stats_ptr, nd["w"] + " stem fwd", produces=y)
            self.profile_meta.append(("fwd", self.fwd[-1], self._kname(d), flops, (x.pixels * 4 + y.pixels * y.c) * self.es))
        if not self.with_bwd:
            return
        # 16-bit runs: all k kernel rows in ONE pass (lh_wgrad_rowfold: dy is read once per input tile, not once per row);
        # the gradient index row*kr + j is gstage's [cout][k][kr] layout, so the fold is a plain sum over the splits
        bo, bi, ns, ring = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        self.lib.lh_wgrad_tile(C.byref(d), y.c, kr, self.dt, C.byref(bo), C.byref(bi), C.byref(ns), C.byref(ring))
        fold = bool(ring.value)
        dw = _desc(x.n, hp, wp, 4, k * kr, y.h, y.w, s, s, y.c, y.h, y.w, 1, 1, 0, 0, y.c, [(0, 0)]) if fold else d
        self.keep.append(dw)
        n_in_w = k * kr if fold else kr
        rs_arr = _taps_array([(0, 0)] if fold else rows)

        def tune_launch(xp, dyp, ws, grad, sp):
            if fold:
                check(self.lib.lh_wgrad_fused(C.byref(dw), k, xp, dyp, y.c, y.c, k * kr, ws, grad, k * kr, 1, 0, 0, rs_arr, 0, self.dt, sp),
                      "autotune lh_wgrad_fused")
            else:
                check(self.lib.lh_wgrad_fused(C.byref(d), 0, xp, dyp, y.c, y.c, kr, ws, grad, k * kr, 1, kr, 0, rs_arr, 0, self.dt, sp),
                      "autotune lh_wgrad_fused")
        self._tune_wgrad(dw, y.c, n_in_w, y.c, tune_launch, y.c * k * kr, tag=("fold", k if fold else 0))
        slab_bytes = self.lib.lh_wgrad_workspace_bytes(C.byref(dw), y.c, n_in_w, self.dt)
        self._ws_wgrad = max(self._ws_wgrad, slab_bytes)
        gstage = self._alloc(cout, k, kr // 4, 4, dtype=torch.float32)
        gw = self.grads[nd["w"] + ".weight"]

        def emit():
            dy = self._act_grad(y)
            if fold:
                a = [C.byref(dw), k, img.data_ptr(), dy.data_ptr(), y.c, y.c, k * kr, 0, gstage.data_ptr(), k * kr, 1, 0, 0, rs_arr, 0, self.dt]
            else:
                a = [C.byref(d), 0, img.data_ptr(), dy.data_ptr(), y.c, y.c, kr, 0, gstage.data_ptr(), k * kr, 1, kr, 0, rs_arr, 0, self.dt]
            tail = 1
            cw = _Call(self.lib.lh_wgrad_fused, None, "stem wgrad", keep=rs_arr, lane=1)

            def set_ws(ptr):
                a[7] = ptr
                cw.args = tuple(a)
            cw.ws_ent = self._ws_note(set_ws, slab_bytes)
            wl = self._wl()
            wl.append(cw)
            self.profile_meta.append(("bwd", wl[-1], self._kname(dw, (y.c, n_in_w)), flops, (x.pixels * 4 + y.pixels * y.c) * self.es))
            wl.append(self._copy4(gw, gstage[:, :, :k, :3].permute(0, 3, 1, 2), "stem grad unstage", lane=tail))
        blk.append(emit)

    # ---- transposed convolution ------------------------------------------------------------------
    def _c_deconv(self, nd, blk):
        x, y, k, p = nd["x"], nd["y"], nd["k"], nd["p"]
        wt = self.params[nd["w"] + ".weight"]          # [cin, cout, k, k]
        cin, cout = wt.shape[0], wt.shape[1]
        xbuf, ybuf = self._act_buf(x), self._act_buf(y)
        all_rs = [(r, q) for r in range(k) for q in range(k)]
        bias = None
        if nd["bias"]:
            bias = self._alloc(y.c, dtype=torch.float32, zero=True)
            bsrc = self.params[nd["bias"]]
            self.packs.append(self._copy4(bias[:cout], bsrc.detach(), "bias pad"))
        descs, packs = [], []
        for ph in range(2):
            for pw in range(2):
                sub = [(r, q) for r, q in all_rs if (ph + p - r) % 2 == 0 and (pw + p - q) % 2 == 0]
                tp = [((ph + p - r) // 2, (pw + p - q) // 2) for r, q in sub]
                gh, gw_ = (y.h - ph + 1) // 2, (y.w - pw + 1) // 2
                descs.append(_desc(x.n, x.h, x.w, x.c, cin, gh, gw_, 1, 1, cout, y.h, y.w, 2, 2, ph, pw, y.c, tp))
                packs.append(self._pack(wt, cout, cin, (k * k, cout * k * k, k, 1), sub, nd["w"] + " deconv pack"))
        with_stats = id(y) in self._bn_inputs and self.training
        if self._phase_rows(descs) > 0:
            self._tune(descs, with_stats=with_stats)
        else:
            for dd in descs:
                self._tune([dd], with_stats=with_stats)
        prow = self._phase_rows(descs)
        if prow > 0:                          # the four sub-pixel phases as ONE launch
            stats = None
            if id(y) in self._bn_inputs and self.training:
                nbytes = self.lib.lh_bn_stats_slab_bytes(4 * prow, y.c)
                y.stats, y.stats_rows = self._alloc((nbytes + 3) // 4, dtype=torch.float32), 4 * prow
                stats = y.stats
            self._igemm_phases(self.fwd, descs, xbuf, packs, ybuf, None, bias, stats, nd["w"] + " deconv fwd", produces=y)
            self.profile_meta.append(("fwd", self.fwd[-1], self._kname(descs[0]), 2.0 * x.pixels * cin * cout * k * k,
                                      (x.pixels * x.c + y.pixels * y.c) * self.es))
            descs = []
        offs = [None] * 4
        if descs and id(y) in self._bn_inputs and self.training:
            offs = self._stats_for(y, descs)
        for d, pk, off in zip(descs, packs, offs):
            st = 0 if off is None else y.stats.data_ptr() + off
            c = self._igemm(self.fwd, d, xbuf, pk, ybuf, None, bias, None, nd["w"] + " deconv fwd", produces=y)
            self._patch(c, stats=st)
            self.profile_meta.append(("fwd", self.fwd[-1], self._kname(d), 2.0 * d.n * d.ho * d.wo * cin * cout * d.ntaps,
                                      (x.pixels * x.c + y.pixels * y.c / 4) * self.es))
        if not self.with_bwd:
            return
        # data gradient = stride-2 convolution of dy; weight gradient gathers dy, dense operand is x
        taps = [(r - p, q - p) for r, q in all_rs]
        dg = _desc(y.n, y.h, y.w, y.c, cout, x.h, x.w, 2, 2, x.c, x.h, x.w, 1, 1, 0, 0, x.c, taps)
        self.keep.append(dg)
        gpack = self._pack(wt, cin, cout, (cout * k * k, k * k, k, 1), all_rs, nd["w"] + " deconv dgrad pack")
        rs_arr = _taps_array(all_rs)

        def tune_launch(xp, dyp, ws, grad, sp):            # dy is the gathered operand here, x the dense one
            check(self.lib.lh_wgrad_fused(C.byref(dg), 0, xp, dyp, x.c, cin, cout, ws, grad, cout * k * k, k * k, k, 1, rs_arr, 0, self.dt, sp),
                  "autotune lh_wgrad_fused")
        self._tune_wgrad(dg, cin, cout, x.c, tune_launch, cin * cout * k * k)
        slab_bytes = self.lib.lh_wgrad_workspace_bytes(C.byref(dg), cin, cout, self.dt)
        self._ws_wgrad = max(self._ws_wgrad, slab_bytes)
        gw = self.grads[nd["w"] + ".weight"]
        flops = 2.0 * x.pixels * cin * cout * k * k

        def emit():
            dy = self._act_grad(y)
            a = [C.byref(dg), 0, dy.data_ptr(), xbuf.data_ptr(), x.c, cin, cout, 0, gw.data_ptr(), cout * k * k, k * k, k, 1, rs_arr, 0, self.dt]
            cw = _Call(self.lib.lh_wgrad_fused, None, nd["w"] + " wgrad", keep=rs_arr, lane=1)
            cw.wargs, cw.wbufs = a, (dy, xbuf)

            def set_ws(ptr):
                a[7] = ptr
                cw.args = tuple(a)
            cw.ws_ent = self._ws_note(set_ws, slab_bytes)
            wl = self._wl()
            wl.append(cw)
            self.profile_meta.append(("bwd", wl[-1], self._kname(dg, (cin, cout)), flops, (x.pixels * x.c + y.pixels * y.c) * self.es))
            if nd["bias"]:
                gb_ = self.grads[nd["bias"]]
                wl.append(self._bias_grad(dy, y, cout, gb_))
            if x.needs_grad:
                for _dd, _nt in self._dgrad([dg], dy, [gpack], x, nd["w"] + " deconv dgrad"):
                    gname, gbytes = self._gate_meta(dg, x)
                    self.profile_meta.append(("bwd", self.bwd[-1], gname or self._kname(dg), flops, gbytes + (x.pixels * x.c + y.pixels * y.c) * self.es))
        blk.append(emit)

    # ---- BatchNorm + sum + ReLU ------------------------------------------------------------------
    def _c_fuse(self, nd, blk):
        terms, out, relu = nd["terms"], nd["out"], nd["relu"]
        obuf = self._act_buf(out)
        c = out.c
        fd = FuseDesc()
        fd.nterms, fd.relu = len(terms), int(relu)
        bn_state = []
        for i, (a, bn, l) in enumerate(terms):
            fd.x[i] = self._act_buf(a).data_ptr()
            fd.log2up[i] = l
            if bn is None:
                bn_state.append(None)
                continue
            st = {k: self._alloc(c, dtype=torch.float32) for k in ("scale", "shift", "mean", "invstd")}
            bn_state.append(st)
            fd.scale[i], fd.shift[i] = st["scale"].data_ptr(), st["shift"].data_ptr()
            P = self.params
            if self.training:
                assert a.stats is not None, f"BN {bn} input has no statistics slab"
                # the finalize (batch statistics -> scale / shift / saved mean, invstd / running statistics) travels WITH the
                # elementwise call (lh_fuse_desc.fin): small tensors run both as one launch, the others launch it first
                fin = _lib.BnFinalizeCall(
                    a.stats.data_ptr(), a.stats_rows, a.pixels, c, P[bn + ".weight"].data_ptr(), P[bn + ".bias"].data_ptr(),
                    P[bn + ".running_mean"].data_ptr(), P[bn + ".running_var"].data_ptr(),
                    _ptr(P.get(bn + ".num_batches_tracked")), BN_MOMENTUM, BN_EPS,
                    st["scale"].data_ptr(), st["shift"].data_ptr(), st["mean"].data_ptr(), st["invstd"].data_ptr())
                self.keep.append(fin)
                fd.fin[i] = C.pointer(fin)
            else:
                self.fwd.append(_Call(self.lib.lh_bn_eval_affine, (
                    P[bn + ".weight"].data_ptr(), P[bn + ".bias"].data_ptr(), P[bn + ".running_mean"].data_ptr(),
                    P[bn + ".running_var"].data_ptr(), BN_EPS, c, st["scale"].data_ptr(), st["shift"].data_ptr()), bn + " eval affine"))
        self.keep.append(fd)
        if not self.training and self._fold_eval_bn(terms, bn_state, out, relu):
            return
        # multi-term ReLU nodes (residual tails, HRNet fuse sums) keep the ReLU mask as one bit per element, so the
        # backward pass reads n*h*w*c/8 bytes instead of the stored activation (single-BN-term nodes recompute the
        # mask from x*scale+shift and need neither)
        relu_bits = None
        if relu and self.with_bwd and len(terms) > 1:
            relu_bits = self._alloc(out.pixels * c // (16 // self.es), dtype=torch.uint8)
            fd.relu_mask = relu_bits.data_ptr()
        if self.training and self.with_bwd and relu and len(terms) == 1 and terms[0][1] is not None and terms[0][2] == 0:
            self._gate_info[id(out)] = dict(raw=terms[0][0], st=bn_state[0])
        bn_terms = [i for i, (_, bn, _) in enumerate(terms) if bn is not None]
        if (self.training and self.with_bwd and relu_bits is not None and len(terms) == 2 and len(bn_terms) == 1
                and all(l == 0 for _, _, l in terms) and all(a.c == out.c for a, _, _ in terms)):
            # a residual tail relu(BN(raw) + identity): its sign is in the mask bits
            self._gate_info[id(out)] = dict(raw=terms[bn_terms[0]][0], st=bn_state[bn_terms[0]], mask=relu_bits)
        if (self.training and self.with_bwd and relu_bits is not None and len(terms) == 2 and len(bn_terms) == 2
                and all(l == 0 for _, _, l in terms) and all(a.c == out.c for a, _, _ in terms)):
            # ... with a projection shortcut: relu(BN(raw) + BN2(raw2))
            self._gate_info[id(out)] = dict(raw=terms[0][0], st=bn_state[0], mask=relu_bits, raw2=terms[1][0], st2=bn_state[1])
        self.fwd.append(_Call(self.lib.lh_fuse_fwd, (C.byref(fd), obuf.data_ptr(), out.n, out.h, out.w, c, self.dt), "fuse fwd"))
        if self.training and relu and len(terms) == 1 and terms[0][1] is not None and terms[0][2] == 0 and relu_bits is None:
            # what a max-pool that follows needs to take this node's elementwise pass over (_c_maxpool)
            self._bnrelu_info[id(out)] = dict(raw=terms[0][0], st=bn_state[0], call=self.fwd[-1], fin=fd.fin[0])
        self.profile_meta.append(("fwd", self.fwd[-1], "fuse_fwd(all kernels)", 0.0, (sum(a.pixels for a, _, _ in terms) + out.pixels) * c * self.es))
        if not self.with_bwd:
            return
        self._ws_fuse = max(self._ws_fuse, self.lib.lh_fuse_bwd_workspace_bytes(out.n, out.h, out.w, c))
        def emit():
            bd = FuseBwdDesc()
            bd.dout = self._act_grad(out).data_ptr()
            bd.out = obuf.data_ptr() if relu else None
            bd.relu_mask = relu_bits.data_ptr() if relu_bits is not None else None
            bd.nterms, bd.relu = len(terms), int(relu)
            bd.strips_cap = 256 if self._emit_group > 1 else 0      # nodes of a batch group share their launches
            pre = self._gated.get(id(out))
            if pre is not None:                                     # dout was written by lh_igemm_gated: gated, with its partial sums
                bd.pre_partial, bd.pre_rows = pre[0].data_ptr(), pre[1]
                if len(pre) > 2 and pre[2] is not None:
                    bd.pre_partial2 = pre[2].data_ptr()
            for i, (a, bn, l) in enumerate(terms):
                bd.log2up[i] = l
                if not a.needs_grad:
                    continue
                if (bn is None and l == 0 and len(terms) == 2 and relu_bits is not None and a.c == out.c
                        and self._nwrites.get(id(a), 0) == 0 and self._next_writer_is_conv(a, nd)):
                    # identity shortcut whose gradient dout * mask would be the first write of a.grad, followed by a
                    # convolution's data gradient: that launch adds it from dout itself (lh_igemm addend_mask)
                    self._masked_addend[id(a)] = (self._act_grad(out), relu_bits)
                    continue
                bd.dx[i] = self._act_grad(a).data_ptr()
                bd.accumulate[i] = 0 if self._first_write(a) else 1
                if bn is not None:
                    st = bn_state[i]
                    bd.x[i] = a.buf.data_ptr()
                    bd.scale[i], bd.save_mean[i], bd.save_invstd[i] = st["scale"].data_ptr(), st["mean"].data_ptr(), st["invstd"].data_ptr()
                    bd.shift[i] = st["shift"].data_ptr()
                    bd.dgamma[i] = self.grads[bn + ".weight"].data_ptr()
                    bd.dbeta[i] = self.grads[bn + ".bias"].data_ptr()
            self.keep.append(bd)
            args = [C.byref(bd), out.n, out.h, out.w, c, 0, self.dt]
            call = _Call(self.lib.lh_fuse_bwd, None, "fuse bwd" + (" (reduce pass done by the gated data gradient)" if pre is not None else ""))

            def set_ws(ptr):
                args[5] = ptr
                call.args = tuple(args)
            self._ws_users_fuse.append((set_ws, self._cur_lane))
            self.bwd.append(call)
            # algorithmic bytes of the BN / ReLU backward of this node: the reduce pass reads dout and every BN term's x, the
            # apply pass reads them again and writes one gradient per term that takes one (5 tensor passes per BN)
            n_bn = sum(1 for _, bn, _ in terms if bn is not None)
            n_dx = sum(1 for i in range(len(terms)) if bd.dx[i])
            # (a node whose dout came from a gated data gradient has no reduce pass: that launch's epilogue read x -- charged to it, _gate_meta)
            passes = ((1 if pre is not None else 2) * (1 + n_bn) if n_bn else 1) + n_dx
            self.profile_meta.append(("bwd", self.bwd[-1], "fuse_bwd(all kernels)", 0.0, float(passes) * out.pixels * c * self.es))
            # what SURVEY 8(d)'s traffic model itself charges to the BatchNorm backward: ONE re-read of y per BatchNorm term
            self.bn_bwd_8d_bytes = getattr(self, "bn_bwd_8d_bytes", 0.0) + float(n_bn) * out.pixels * c * self.es
        blk.append(emit)

    def _next_writer_is_conv(self, a, nd):
        """True when, walking backward from fuse node `nd`, the next writer of a.grad is a stride-1-output convolution /
        transposed convolution data gradient over a dense tensor (it can take a masked addend)."""
        uses = self._uses.get(id(a), [])
        idx = [i for i, (_, n) in enumerate(uses) if n is nd]
        if len(idx) != 1 or idx[0] == 0:
            return False
        kind, nxt = uses[idx[0] - 1]
        return kind in ("conv", "deconv") and a.c == a.c_valid


    def _c_maxpool(self, nd, blk):
        x, y = nd["x"], nd["y"]
        if self._fuse_stem_pool(nd):
            return
        xbuf, ybuf = self._act_buf(x), self._act_buf(y)
        idx = self._alloc(y.n, y.h, y.w, y.c, dtype=torch.uint8) if self.with_bwd else None     # window positions: only the backward pass reads them
        bi = self._bnrelu_info.get(id(x)) if os.environ.get("LH_BN_POOL", "1") != "0" else None
        nchunk = x.c * self.es // 16
        if bi is not None and len(self._uses.get(id(x), [])) == 1 and self.fwd and self.fwd[-1] is bi["call"] and x.c == x.c_valid \
                and nchunk & (nchunk - 1) == 0 and nchunk <= 256:            # (the flat BN-backward kernels: they take the mask from raw)
            # x = relu(BN(raw)) feeds this pool alone (the training stem, pose_resnet.py:153-156): the pool reads RAW, applies the
            # BatchNorm affine + ReLU per tap (rounded as the stored activation would be: bit-identical pooled values and
            # positions) and x -- the largest activation of the network -- is never written: the backward pass works from
            # idx (pool) and recomputes the ReLU mask from raw (lh_fuse_bwd), it never reads x
            self.fwd.pop()
            self.profile_meta = [m for m in self.profile_meta if m[1] is not bi["call"]]
            st = bi["st"]
            if bi["fin"]:                                           # the finalize that travelled with the elementwise call: a launch of its own
                arr = (_lib.BnFinalizeCall * 1)(bi["fin"].contents)
                self.keep.append(arr)
                self.fwd.append(_Call(self.lib.lh_bn_finalize_multi, (arr, 1), "bn finalize"))
            self.fwd.append(_Call(self.lib.lh_bn_relu_maxpool3x3s2_fwd, (bi["raw"].buf.data_ptr(), st["scale"].data_ptr(), st["shift"].data_ptr(),
                                                                          ybuf.data_ptr(), _ptr(idx), x.n, x.h, x.w, x.c, self.dt), "bn + relu + maxpool fwd"))
            self.profile_meta.append(("fwd", self.fwd[-1], "maxpool_fwd_kernel(bn)", 0.0, (x.pixels * x.c + y.pixels * y.c) * self.es))
        else:
            self.fwd.append(_Call(self.lib.lh_maxpool3x3s2_fwd, (xbuf.data_ptr(), ybuf.data_ptr(), _ptr(idx), x.n, x.h, x.w, x.c, self.dt), "maxpool fwd"))
        if not self.with_bwd:
            return

        def emit():
            dy, dx = self._act_grad(y), self._act_grad(x)
            assert self._first_write(x), "maxpool input gradient must be produced by the pool alone"
            gi = self._gate_info.get(id(x)) if (self.bn_gate and self.n_lanes == 1 and self.es == 2 and os.environ.get("LH_POOL_GATE", "1") != "0") else None
            nch = x.c * self.es // 16
            if gi is not None and gi.get("mask") is None and len(self._uses.get(id(x), [])) == 1 and x.c == x.c_valid and nch & (nch - 1) == 0 and nch <= 256:
                # x = relu(BN(raw)) with the pool as its only reader: the pool's backward stores the ReLU-gated gradient and the
                # BatchNorm-backward partial sums (the node's backward skips its reduce pass), as lh_igemm_gated does for convolutions
                rows = self.lib.lh_maxpool3x3s2_bwd_gated_rows(x.n, x.h, x.w, x.c, self.dt)
                partial = self._alloc(rows * 2 * x.c, dtype=torch.float32)
                st = gi["st"]
                gate = _lib.BnBwdGate(gi["raw"].buf.data_ptr(), st["mean"].data_ptr(), st["invstd"].data_ptr(), st["scale"].data_ptr(),
                                      st["shift"].data_ptr(), partial.data_ptr())
                self.keep.append(gate)
                self.bwd.append(_Call(self.lib.lh_maxpool3x3s2_bwd_gated, (dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), C.byref(gate), x.n, x.h, x.w, x.c,
                                                                         self.dt), "maxpool bwd + BN-backward gate"))
                self._gated[id(x)] = (partial, rows)
            else:
                self.bwd.append(_Call(self.lib.lh_maxpool3x3s2_bwd, (dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), x.n, x.h, x.w, x.c, self.dt), "maxpool bwd"))
        blk.append(emit)

    def use_uint8_input(self, hs, ws, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), jitter=False):
        """Switch the plan's input to raw uint8 HWC images [n][hs][ws][3]: ToTensor + bilinear Resize [+ ColorJitter] +
        Normalize (the reference's CPU transform chain, src/tools/dataset.py:128-159, defaults = its ImageNet
        constants) run fused on the device and write the stem's padded NHWC4 input.  With ``jitter`` the plan owns
        ``jitter_factors`` (fp32 [n][4]: brightness, contrast, saturation, hue) and ``jitter_order`` (int32 [n][4]: op
        ids, negative = skip) that the caller fills per batch (``sample_color_jitter``).  Returns the static uint8
        input buffer."""
        self.img_u8 = self._alloc(self.n, hs, ws, 3, dtype=torch.uint8, zero=True)
        m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
        self.keep += [m3, s3]
        if jitter:
            self.jitter_factors = self._alloc(self.n, 4, dtype=torch.float32, zero=True)
            self.jitter_factors[:, :3] = 1.0                                    # identity until the caller draws
            self.jitter_order = torch.full((self.n, 4), -1, dtype=torch.int32, device=self.device)
            ws_j = self._alloc(self.lib.lh_image_jitter_workspace_bytes(self.n), dtype=torch.uint8)
            self.keep.append(self.jitter_order)
            self.fwd[self._image_call_index] = _Call(self.lib.lh_image_u8_jitter_to_nhwc4, (
                self.img_u8.data_ptr(), self.img_nhwc4.data_ptr(), self.n, hs, ws, self.h, self.w, self.img_pad, self.img_wp,
                m3, s3, self.jitter_factors.data_ptr(), self.jitter_order.data_ptr(), ws_j.data_ptr(), self.dt),
                "uint8 input pipeline + ColorJitter")
            return self.img_u8
        self.fwd[self._image_call_index] = _Call(self.lib.lh_image_u8_to_nhwc4, (
            self.img_u8.data_ptr(), self.img_nhwc4.data_ptr(), self.n, hs, ws, self.h, self.w, self.img_pad, self.img_wp,
            m3, s3, self.dt), "uint8 input pipeline")
        return self.img_u8

    # ------------------------------------------------------------------ run
    def refresh_packs(self, stream, overlap=False, side_work=None):
        """Rebuild the device-side weight packs from the parameter arena.  overlap=True (the captured training step): the
        one large launch -- the tiled transposing pack of every regular convolution, ~0.12 ms -- runs on a side stream
        under the image transform, the stem and the pool; the forward list waits for it at its 'packjoin' marker, just
        before the first launch that reads a regular pack.  side_work(stream): more work for that side stream that only
        depends on the step's inputs (the target render); returns True when it was run there."""
        side = getattr(self, "_pack_stream", None)
        if not overlap or side is None or self._packjoin_at is None:
            for c in self.packs:
                c(stream)
            return False
        main = torch.cuda.current_stream()
        assert main.cuda_stream == stream
        side.wait_event(main.record_event())
        self._pack_late = None
        for c in self.packs:
            if c.fn is self.lib.lh_pack_weights_tiled:
                if c.lane == 1 and self._late_packs is not None:
                    self._pack_late = c          # launched when the forward list reaches its 'packfork2' marker
                else:
                    c(side.cuda_stream)
            else:
                c(stream)
        if side_work is not None:
            side_work(side.cuda_stream)
        self._pack_event = side.record_event()
        return side_work is not None

    def _run_lanes(self, calls, stream, hooks=None):
        """Launch `calls` with the independent branch chains (stream lane > 0) on side streams: a lane's first launch
        after a fork waits for the fork's event on the main stream, the join makes the main stream wait for every lane
        used since; outside fork/join regions (and at the end of the slice) everything is ordered on the main stream.
        Works eagerly and under hipGraph capture (the side streams join the capture through the events).
        hooks: {i: fn(events)} -- before calls[i] is launched, fn receives events that cover everything launched so far
        (main stream + every side stream used): work that only needs calls[:i] hangs off them without stalling any lane."""
        main = torch.cuda.current_stream()
        assert main.cuda_stream == stream, "lanes need the launch stream to be torch's current stream"
        ev, forked, used = None, set(), set()
        wev, wused = {}, set()                 # weight-gradient side streams: pending event per stream, streams used
        for ci, c in enumerate(calls):
            if hooks and ci in hooks:
                hooks[ci]([main.record_event()] + [self._lane_streams[L].record_event() for L in sorted(used | wused)])
            if isinstance(c, _Marker):
                if c.kind == "packjoin":
                    if self._pack_event is not None:
                        main.wait_event(self._pack_event)
                        self._pack_event = None
                elif c.kind == "packfork2":      # the late pack group starts here, on the pack stream, under the launches that follow
                    if self._pack_late is not None:
                        self._pack_stream.wait_event(main.record_event())
                        self._pack_late(self._pack_stream.cuda_stream)
                        self._pack_event2 = self._pack_stream.record_event()
                        self._pack_late = None
                elif c.kind == "packjoin2":
                    if self._pack_event2 is not None:
                        main.wait_event(self._pack_event2)
                        self._pack_event2 = None
                elif c.kind == "wfork":          # the deferred weight gradients that follow may start once their source
                    src = main if c.lane == 0 else self._lane_streams[c.lane]      # stream got here
                    wev[c.slane] = src.record_event()
                elif c.kind == "fork":
                    ev, forked = main.record_event(), set()
                else:
                    for L in used:
                        main.wait_stream(self._lane_streams[L])
                    ev, used = None, set()
                continue
            L = c.slane
            if L == 0:
                c(stream)
                continue
            if L < 0:                          # deferred weight-gradient group
                s = self._lane_streams[L]
                e = wev.pop(L, None)
                if e is not None:
                    s.wait_event(e)
                elif L not in wused:
                    s.wait_stream(main)        # slice starts inside a group (data-parallel segments)
                wused.add(L)
                c(s.cuda_stream)
                continue
            s = self._lane_streams.get(L)
            if s is None:
                s = self._lane_streams[L] = torch.cuda.Stream()
            if L not in forked:
                if ev is not None:
                    s.wait_event(ev)
                else:
                    s.wait_stream(main)
                forked.add(L)
            used.add(L)
            c(s.cuda_stream)
        for L in used | wused:
            main.wait_stream(self._lane_streams[L])

    def run_forward(self, stream):
        if self.use_lanes:
            return self._run_lanes(self.fwd, stream)
        for c in self.fwd:
            if not isinstance(c, _Marker):
                c(stream)
            elif c.kind == "packjoin" and self._pack_event is not None:
                torch.cuda.current_stream().wait_event(self._pack_event)
                self._pack_event = None
            elif c.kind == "packfork2" and self._pack_late is not None:
                self._pack_late(stream)              # no side streams in this plan: the late group runs in place
                self._pack_late = None

    def run_backward(self, stream, lo=0, hi=None, hooks=None):
        """Run bwd[lo:hi] (a segment of the backward list: data-parallel plans replay it bucket by bucket).
        hooks: {index in the backward list: fn(events)}, called when everything before that index has been launched, with
        events that cover it (TrainStep: the Adam update of the parameters whose gradients are final by then)."""
        calls = self.bwd[lo:hi]
        hooks = {i - lo: f for i, f in hooks.items() if lo <= i < (len(self.bwd) if hi is None else hi)} if hooks else None
        if self.use_lanes:
            return self._run_lanes(calls, stream, hooks)
        for ci, c in enumerate(calls):
            if hooks and ci in hooks:
                hooks[ci]([torch.cuda.current_stream().record_event()])
            if not isinstance(c, _Marker):
                c(stream)

    def forward(self, images, repack=True):
        """images: fp32 NCHW on the device.  Returns the plan's fp32 NCHW heatmap buffer."""
        if tuple(images.shape) != (self.n, 3, self.h, self.w):
            raise _lib.LightHandError(f"plan was built for {(self.n, 3, self.h, self.w)}, got {tuple(images.shape)}")
        self.img_nchw.copy_(images)
        stream = torch.cuda.current_stream().cuda_stream
        if repack:
            self.refresh_packs(stream)
        self.run_forward(stream)
        return self.out_nchw

    def backward(self, dheat):
        self.dout_nchw.copy_(dheat)
        self.run_backward(torch.cuda.current_stream().cuda_stream)
