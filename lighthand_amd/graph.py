"""What a model's description is made of and what a plan is a list of: activations, the graph builder a model's ``describe`` talks to,
convolution descriptors, pre-bound C-ABI calls and the fork / join markers of the stream lanes.  (Split out of engine.py in round 6;
``lighthand_amd.engine`` re-exports every name.)"""
import ctypes as C

import torch

from ._lib import IgemmDesc, check

PRECISIONS = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}
BN_MOMENTUM = 0.1   # src/modeling/simplebaseline/pose_resnet.py:19, src/modeling/hrnet/pose_hrnet.py:18
BN_EPS = 1e-5


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class Act:
    """One NHWC activation (and, in training plans, its gradient)."""
    __slots__ = ("n", "h", "w", "c", "c_valid", "buf", "grad", "needs_grad", "stats", "stats_rows",
                 "is_image", "name")

    def __init__(self, n, h, w, c, c_valid=None, name=""):
        self.n, self.h, self.w, self.c = n, h, w, c
        self.c_valid = c if c_valid is None else c_valid
        self.buf = self.grad = self.stats = None
        self.stats_rows = 0
        self.needs_grad = True
        self.is_image = False
        self.name = name

    @property
    def pixels(self):
        return self.n * self.h * self.w


# --------------------------------------------------------------------------------------- graph
class GraphBuilder:
    """Collects the nodes a model emits from ``describe``.  Parameter names are state_dict keys."""

    def __init__(self, n, h, w, params):
        self.n, self.h, self.w = n, h, w
        self.params = params
        self.nodes = []
        self.node_lanes = []          # stream lane of every node (0 = main); see fork() / join()
        self.lane = 0
        self.out = None

    def _add(self, kind, nd):
        self.nodes.append((kind, nd))
        self.node_lanes.append(self.lane)

    def fork(self):
        """Nodes described between fork() and join() with ``gb.lane = i`` (i > 0) form chains that are independent of the
        other lanes' chains (HRNet's parallel branches): the plan may run them on separate HIP streams."""
        self._add("fork", {})

    def join(self):
        self.lane = 0
        self._add("join", {})

    def input(self):
        a = Act(self.n, self.h, self.w, 3, name="input")
        a.is_image = True
        a.needs_grad = False
        self._add("input", a)
        return a

    def input_act(self, c, h=None, w=None):
        """A dense NHWC activation fed directly (kernel tests, sub-networks); it takes gradients."""
        a = Act(self.n, h or self.h, w or self.w, c, name="input_act")
        self._add("input_act", a)
        return a

    def conv(self, x, wname, k, stride, pad, bias=None):
        w = self.params[wname + ".weight"]
        cout, cin = w.shape[0], w.shape[1]
        assert w.shape[2] == k and w.shape[3] == k, wname
        assert x.is_image or cin == x.c_valid, (wname, cin, x.c_valid)
        ho = (x.h + 2 * pad - k) // stride + 1
        wo = (x.w + 2 * pad - k) // stride + 1
        y = Act(x.n, ho, wo, (cout + 31) // 32 * 32 if cout % 8 else cout, cout, name=wname)
        self._add("conv", dict(x=x, y=y, w=wname, k=k, s=stride, p=pad, bias=bias))
        return y

    def deconv(self, x, wname, k, bias=None):
        w = self.params[wname + ".weight"]          # [C_in, C_out, k, k]
        assert w.shape[0] == x.c_valid and w.shape[2] == k
        pad, opad = {4: (1, 0), 3: (1, 1), 2: (0, 0)}[k]     # pose_resnet.py:194-205
        ho = (x.h - 1) * 2 - 2 * pad + k + opad
        wo = (x.w - 1) * 2 - 2 * pad + k + opad
        y = Act(x.n, ho, wo, w.shape[1], name=wname)
        self._add("deconv", dict(x=x, y=y, w=wname, k=k, p=pad, bias=bias))
        return y

    def fuse(self, terms, relu=True):
        """terms: Act (identity) | (Act, bn_prefix) | (Act, bn_prefix, log2_upsample)."""
        norm = []
        for t in terms:
            if isinstance(t, Act):
                norm.append((t, None, 0))
            elif len(t) == 2:
                norm.append((t[0], t[1], 0))
            else:
                norm.append(tuple(t))
        base = max(norm, key=lambda t: t[0].h << t[2])
        h, w = base[0].h << base[2], base[0].w << base[2]
        for a, _, l in norm:
            assert (a.h << l, a.w << l) == (h, w) and a.c == norm[0][0].c
        out = Act(norm[0][0].n, h, w, norm[0][0].c, name="fuse")
        self._add("fuse", dict(terms=norm, out=out, relu=relu))
        return out

    def maxpool(self, x):
        y = Act(x.n, (x.h + 2 - 3) // 2 + 1, (x.w + 2 - 3) // 2 + 1, x.c, name="maxpool")
        self._add("maxpool", dict(x=x, y=y))
        return y

    def output(self, y):
        self.out = y
        self._add("output", dict(y=y))


def _desc(n, hi, wi, pix_stride, k_run, ho, wo, sh, sw, cout, OH, OW, osh, osw, ooh, oow, out_stride, taps):
    d = IgemmDesc()
    d.n, d.hi, d.wi, d.in_pix_stride, d.k_run = n, hi, wi, pix_stride, k_run
    d.ho, d.wo, d.sh, d.sw, d.cout = ho, wo, sh, sw, cout
    d.OH, d.OW, d.osh, d.osw, d.ooh, d.oow, d.out_pix_stride = OH, OW, osh, osw, ooh, oow, out_stride
    d.ntaps, d.relu = len(taps), 0
    assert len(taps) <= 64
    for i, (dh, dw) in enumerate(taps):
        assert -128 <= dh < 128 and -128 <= dw < 128
        d.dh[i], d.dw[i] = dh, dw
    return d


def _taps_array(rs):
    flat = [v for t in rs for v in t] or [0, 0]
    return (C.c_int * len(flat))(*flat)


class _Call:
    """A pre-bound C-ABI call; the stream is appended at run time.  ``lane`` 1 marks work that may run on
    the side stream of the backward pass (weight gradients: they only feed the optimizer)."""
    __slots__ = ("fn", "args", "what", "keep", "lane", "ig", "slane", "mtag", "keep_desc", "ws_ent", "wargs", "wbufs")

    def __init__(self, fn, args, what, keep=None, lane=0):
        self.fn, self.args, self.what, self.keep, self.lane = fn, args, what, keep, lane
        self.ig = None               # argument positions for Plan._patch (lh_igemm layout unless set)
        self.slane = 0               # stream lane (branch) the call belongs to
        self.keep_desc = None        # weight-gradient calls: their descriptor (Plan._batch_wgrads)
        self.mtag = None             # (group, section, member, position): calls of one batch group that may merge (Plan._merge_groups)
        self.ws_ent = None           # weight-gradient calls: their entry in Plan._ws_users (the slab follows the call's stream)
        self.wargs = None            # weight-gradient calls: the argument list of lh_wgrad_fused (Plan._table_wgrads reads it before the slab is bound)
        self.wbufs = None            # ... and the tensors behind its x / dy pointers

    def __call__(self, stream):
        rc = self.fn(*self.args, stream)
        if rc:
            check(rc, self.what)


class _Marker:
    """fork / join point of the stream lanes inside a launch list."""
    __slots__ = ("kind", "what", "lane", "slane")

    def __init__(self, kind):
        self.kind, self.what, self.lane, self.slane = kind, kind, 0, 0

    def __call__(self, stream):          # a plain in-order replay of a launch list (profilers) just skips it
        return None
