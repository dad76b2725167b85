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


# --------------------------------------------------------------------------------------- plan
class Plan:
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
        self.n_lanes = max(gb.node_lanes) + 1 if gb.node_lanes else 1
        self.use_lanes = self.n_lanes > 1
        # Multi-problem launches: the nodes at the same position of the parallel chains of a fork .. join region (HRNet's
        # branches) are compiled as a GROUP whose launches merge into lh_*_multi calls (one grid for 2-4 problems) on the
        # main stream, instead of one launch per branch on stream lanes.  16-bit types; LH_BATCH=0 keeps the lanes.
        self.batch = os.environ.get("LH_BATCH", "1") != "0" and self.es == 2 and self.n_lanes > 1
        self.batch_split = os.environ.get("LH_BATCH", "1") == "2"
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

    # ------------------------------------------------------------------ kernel autotuning
    _TUNE_CACHE = {}        # launch signature -> (bm, bp, depth, kb): shared by every plan of the process
    # test hooks: force_cfg(candidates) -> (bm, bp, depth, kb) | None and force_wgrad(candidates) -> (bo, bi, enc) | None
    # replace the measurement for the plans built while they are set (tests walk every compiled-in configuration)
    force_cfg = None
    force_wgrad = None
    fuse_head = True             # test hook: False keeps the inference head as separate launches (A/B against _fuse_head)
    fuse_stem = True             # test hook: False keeps the inference stem as convolution + max-pool launches (A/B against _fuse_stem_pool)
    fuse_bottleneck = os.environ.get("LH_FUSE_BOTTLENECK", "1") != "0"    # False keeps the stage-1 bottlenecks of inference plans as three launches (A/B against _fuse_bottleneck)
    _tune_file_loaded = False

    _tune_measured = set()      # keys measured by this process or read from the user's cache file (what a save writes)

    @staticmethod
    def _tune_cache_path():
        """Where measured choices persist.  LH_TUNE_CACHE=<file> names it, LH_TUNE_CACHE=0 turns persistence off; default
        ON at $XDG_CACHE_HOME/lighthand_amd/tune_gfx950.txt: the weight gradient's pixel-split count (fp32 summation
        order) and the forward tile (number of BN partial-sum rows) are measured choices, so a restarted or resumed job
        must start from the SAME choices to reproduce its sums bit for bit (timing noise may flip a near-tie)."""
        path = os.environ.get("LH_TUNE_CACHE")
        if path in ("0", "off", "none"):
            return None
        if not path:
            base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
            path = os.path.join(base, "lighthand_amd", "tune_gfx950.txt")
        return path

    @staticmethod
    def _lib_stamp():
        """Identifies the build of the kernel library the choices were measured with (size + modification time of the
        .so): a cache file written by another build is ignored, its choices may name kernels that no longer exist or no
        longer win."""
        try:
            st = os.stat(_lib.LIB_PATH)
            return "lib %d %d" % (st.st_size, int(st.st_mtime))
        except OSError:
            return "lib ?"

    @staticmethod
    def _parse_tune_line(line):
        """(key, value) of one line of a tuning file, or None for a line that does not parse (a truncated write, an edit)."""
        import ast
        try:
            k, v = ast.literal_eval(line)
            return k, tuple(v)
        except (ValueError, SyntaxError, TypeError):
            return None

    @classmethod
    def _tune_cache_io(cls, save=False):
        """Measured choices persist across processes (a restarted job, or a profiling run that should not contain the
        tuner's own launches, starts from the file; new measurements are written back).  Precedence: the user's file
        (local measurements, only when written by THIS build of the library) over the shipped database; a save writes
        only what was measured locally.  Unparsable lines are skipped; every entry is validated against the compiled-in
        candidates where it is used (a stale one is measured again)."""
        path = cls._tune_cache_path()
        if save:
            if not path:
                return
            try:
                os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
                tmp = path + ".tmp%d" % os.getpid()
                with open(tmp, "w") as f:
                    f.write("# " + cls._lib_stamp() + "\n")
                    for k in cls._tune_measured:
                        if k in cls._TUNE_CACHE:
                            f.write(repr((k, cls._TUNE_CACHE[k])) + "\n")
                os.replace(tmp, path)
            except OSError:
                pass                                  # read-only home: the choices still hold for this process
            return
        if cls._tune_file_loaded:
            return
        cls._tune_file_loaded = True
        if path and os.path.isfile(path):
            lines = open(path).read().splitlines()
            if lines and lines[0].strip() == "# " + cls._lib_stamp():
                for line in lines[1:]:
                    kv = cls._parse_tune_line(line) if line.strip() and not line.startswith("#") else None
                    if kv is not None:
                        cls._TUNE_CACHE[kv[0]] = kv[1]
                        cls._tune_measured.add(kv[0])
        # the shipped database: choices measured on MI355X for the benchmark configurations (tools/make_tune_db.sh);
        # entries are validated against the compiled-in configurations when used, anything else is measured on the fly
        sw = os.environ.get("LH_TUNE_DB", "1")                 # 0 = ignore the database, a path = use that file instead (experiments)
        db = sw if sw not in ("0", "1") else os.path.join(os.path.dirname(os.path.abspath(__file__)), "tune_db_gfx950.txt")
        if os.path.isfile(db) and sw != "0":
            for line in open(db):
                kv = cls._parse_tune_line(line) if line.strip() and not line.startswith("#") else None
                if kv is not None:
                    cls._TUNE_CACHE.setdefault(kv[0], kv[1])

    @staticmethod
    def _desc_key(d):
        return (d.n, d.hi, d.wi, d.in_pix_stride, d.k_run, d.ho, d.wo, d.sh, d.sw, d.cout, d.OH, d.OW, d.osh, d.osw,
                d.ooh, d.oow, d.out_pix_stride, d.ntaps, bytes(d.dh)[:d.ntaps], bytes(d.dw)[:d.ntaps])

    @staticmethod
    def tune_iters():
        """Timed launches per candidate configuration: 4 at plan build (tuning must stay cheap), more when the shipped
        database is generated (LH_TUNE_ITERS, tools/make_tune_db.sh: a 20-launch average ranks near-ties reliably)."""
        return max(1, int(os.environ.get("LH_TUNE_ITERS", "4")))

    _flush_buf = {}

    def _timed_cold(self, run, warm, iters):
        """Time `iters` launches of run() one at a time in the cache state the launch meets inside a step: the caches are
        flushed (a 512 MiB fill, larger than the Infinity Cache), then the operands in `warm` -- tensors the preceding
        kernel of the step has just WRITTEN -- are rewritten from a twin copy, which leaves them in L2 / Infinity Cache
        the way a producer does.  Back-to-back launches on the same scratch operands (LH_TUNE_COLD=0) re-read everything
        from the caches and rank the configurations of the streaming layers wrongly: measured on the 1x1 layers of stage
        1, 28 vs 29 us back to back but 57 vs 67 us cold (tools/pw_bench.py).  Returns milliseconds for all launches."""
        dev = self.device
        fb = Plan._flush_buf.get(dev)
        if fb is None:
            fb = Plan._flush_buf[dev] = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        twins = [(t, self._scratch("twin%d" % i, t.numel() * t.element_size(), like=t)) for i, t in enumerate(warm)]
        stream = torch.cuda.current_stream()
        evs = []
        for _ in range(iters):
            fb.zero_()
            for t, tw in twins:
                t.copy_(tw)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            run()
            b.record(stream)
            evs.append((a, b))
        evs[-1][1].synchronize()
        return sum(a.elapsed_time(b) for a, b in evs)

    def _scratch(self, name, nbytes, like=None):
        if like is not None:                      # a twin of `like`: same bytes, kept for the producer-emulating rewrite
            t = self._tune_bufs.get(name)
            if t is None or t.numel() != like.numel() or t.dtype != like.dtype:
                t = like.clone()
                self._tune_bufs[name] = t
            else:
                t.copy_(like)
            return t
        t = self._tune_bufs.get(name)
        if t is None or t.numel() < nbytes:
            t = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
            if name not in ("out", "wws"):
                # random bit patterns in every operand: all-zero operands let the chip clock higher and would rank the
                # MFMA-heavy configurations too well (cdna guide, methodology rule 25)
                t.view(torch.int16).random_(-16000, 16000) if self.es == 2 else t.view(torch.float32).normal_()
            self._tune_bufs[name] = t
        return t

    _MAX_CANDS = 320          # one size for every lh_igemm_candidates buffer (the 16-bit table holds ~60 entries per launch)

    def _igemm_candidates(self, desc):
        """(buffer of 5 ints per candidate, count) of the configurations compiled in for this launch; a list that fills the
        buffer would have been cut short silently, so that is an error."""
        buf = (C.c_int * (5 * Plan._MAX_CANDS))()
        n = self.lib.lh_igemm_candidates(C.byref(desc), self.dt, buf, Plan._MAX_CANDS)
        if not 0 <= n < Plan._MAX_CANDS:
            raise _lib.LightHandError(f"lh_igemm_candidates returned {n} entries for a buffer of {Plan._MAX_CANDS}")
        return buf, n

    def _tune(self, descs, with_stats=False, addend=None, role=None):
        """Measured kernel choice (cdna guide: measure, don't guess): time every compiled-in configuration that fits
        this launch (lh_igemm_candidates) on scratch operands of the real size and write the fastest into the
        descriptors' cfg.  One descriptor = lh_igemm; several = the phases of lh_igemm_phases (one shared choice).
        Results do not depend on the choice (the K-loop order is the same for every tile).  LH_AUTOTUNE=0 keeps the
        library's static default."""
        if os.environ.get("LH_AUTOTUNE", "1") == "0":
            return
        if self._forced is not None and role in self._forced and len(descs) == 1:     # member of a batch group: the group's choice
            choice = self._forced[role]
            if isinstance(choice, list):              # mixed launch: a configuration per member (direct 3x3 | the shared tile)
                choice = choice[self._forced["member"]]
            for d in descs:
                d.cfg[0], d.cfg[1], d.cfg[2], d.cfg[3] = choice
            return
        lead = max(descs, key=lambda d: d.ntaps)
        if lead.ntaps == 0:
            return
        # addend: None | 'plain' | 'masked' -- the epilogue of an accumulating / masked-addend data gradient moves up to three
        # times the bytes of a plain one, which shifts the best tile
        key = (self.dt, with_stats) + tuple(self._desc_key(d) for d in descs) + ((addend,) if addend else ())
        hit = Plan._TUNE_CACHE.get(key) if Plan.force_cfg is None else None
        buf, n = self._igemm_candidates(lead)
        cands = [tuple(buf[5 * i:5 * i + 4]) for i in range(n)]
        if len(descs) > 1:
            cands = [c for c in cands if c[2] not in (1, 100)]  # the persistent kernels take single launches only
        if len(descs) > 1 and self._phase_rows(descs) <= 0:
            cands = []                                          # phases that cannot be batched: keep the default
        if hit is not None and hit != (0, 0, 0, 0) and hit not in cands:
            hit = None                                          # stale entry (configuration no longer compiled in): measure again
        if hit is None:
            if Plan.force_cfg is not None:
                forced = Plan.force_cfg(cands) if cands else None
                for d in descs:
                    d.cfg[0], d.cfg[1], d.cfg[2], d.cfg[3] = forced or (0, 0, 0, 0)
                return
            hit = (0, 0, 0, 0)
            if len(cands) > 1:
                es = self.es
                kpad = (lead.k_run * es + 127) // 128 * 128
                src = self._scratch("in", lead.n * lead.hi * lead.wi * lead.in_pix_stride * es + 256)
                dst = self._scratch("out", lead.n * lead.OH * lead.OW * lead.out_pix_stride * es + 256)
                packs = [self._scratch(f"pack{i}", (d.cout + 255) // 256 * 256 * max(d.ntaps, 1) * kpad + 256) for i, d in enumerate(descs)]
                rows = max((lead.n * lead.ho * lead.wo + 63) // 64, 1024) * len(descs)     # pointwise candidates: one row per workgroup
                stats = self._scratch("stats", rows * 2 * lead.cout * 4 + 256) if with_stats else None
                dense = lead.out_pix_stride == lead.cout
                add = self._scratch("addend", lead.n * lead.OH * lead.OW * lead.out_pix_stride * es + 256) if addend else None
                amask = self._scratch("amask", lead.n * lead.OH * lead.OW * lead.out_pix_stride * es // 16 + 256) if addend == "masked" and dense else None
                stream = torch.cuda.current_stream()
                sp = stream.cuda_stream
                if len(descs) > 1:
                    arr = (C.POINTER(IgemmDesc) * len(descs))(*[C.pointer(d) for d in descs])
                    parr = (C.c_void_p * len(descs))(*[pk.data_ptr() for pk in packs])

                    def run():
                        check(self.lib.lh_igemm_phases(arr, len(descs), src.data_ptr(), parr, dst.data_ptr(), _ptr(add), _ptr(amask), None, None, None,
                                                       _ptr(stats), self.dt, sp), "autotune lh_igemm_phases")
                else:
                    def run():
                        check(self.lib.lh_igemm(C.byref(lead), src.data_ptr(), packs[0].data_ptr(), dst.data_ptr(), _ptr(add), _ptr(amask), None, None, None,
                                                _ptr(stats), self.dt, sp), "autotune lh_igemm")
                best = None
                cold = os.environ.get("LH_TUNE_COLD", "1") != "0"
                for cfg in cands:
                    for d in descs:
                        d.cfg[0], d.cfg[1], d.cfg[2], d.cfg[3] = cfg
                    run()
                    if cold:                  # input = the previous kernel's output (warm), everything else cold
                        t = self._timed_cold(run, [src[:lead.n * lead.hi * lead.wi * lead.in_pix_stride * es]], Plan.tune_iters())
                    else:
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record(stream)
                        for _ in range(Plan.tune_iters()):
                            run()
                        b.record(stream)
                        b.synchronize()
                        t = a.elapsed_time(b)
                    if os.environ.get("LH_TUNE_LOG"):
                        print(f"[tune {role or ''} {lead.k_run}x{lead.ntaps}->{lead.cout} M={lead.n * lead.ho * lead.wo} addend={addend}] cfg {cfg}: "
                              f"{t / Plan.tune_iters() * 1e3:7.1f} us", flush=True)
                    if best is None or t < best[0]:
                        best = (t, cfg)
                hit = best[1]
            Plan._TUNE_CACHE[key] = hit
            Plan._tune_measured.add(key)
        for d in descs:
            d.cfg[0], d.cfg[1], d.cfg[2], d.cfg[3] = hit

    def _tune_wgrad(self, d, n_out, n_in, dy_stride, launch, grad_floats, tag=()):
        """Measured plan of one weight-gradient launch + its fold: (tile, stage rows, ring depth, pixel splits) from
        lh_wgrad_candidates, timed on scratch operands; the winner goes into d.cfg[5..7].  The split count changes the
        fp32 summation order (deterministically): the choice is cached per launch signature for the whole process so
        that every plan of a process computes the same sums."""
        if os.environ.get("LH_AUTOTUNE", "1") == "0":
            return
        if self._forced is not None and "wgrad" in self._forced and not tag:
            d.cfg[5], d.cfg[6], d.cfg[7] = self._forced["wgrad"][self._forced["member"]]
            return
        key = ("w", self.dt, self._desc_key(d), n_out, n_in, dy_stride) + tuple(tag)
        hit = Plan._TUNE_CACHE.get(key) if Plan.force_wgrad is None else None
        buf = (C.c_int * (5 * 320))()
        n = self.lib.lh_wgrad_candidates(C.byref(d), n_out, n_in, self.dt, buf, 320)
        cands = [tuple(buf[5 * i:5 * i + 5]) for i in range(n)]
        if hit is not None and hit != (0, 0, 0) and hit not in [c[:3] for c in cands]:
            hit = None
        if hit is None:
            if Plan.force_wgrad is not None:
                d.cfg[5], d.cfg[6], d.cfg[7] = (Plan.force_wgrad(cands) if cands else None) or (0, 0, 0)
                return
            hit = (0, 0, 0)
            if len(cands) > 1:
                es = self.es
                xs = self._scratch("in", d.n * d.hi * d.wi * d.in_pix_stride * es + 256)
                dys = self._scratch("dy", d.n * d.ho * d.wo * dy_stride * es + 256)
                slab = self._scratch("wws", (max(c[4] for c in cands) + 1) << 20)
                grad = self._scratch("stats", grad_floats * 4 + 256)
                stream = torch.cuda.current_stream()
                sp = stream.cuda_stream
                best = None
                cold = os.environ.get("LH_TUNE_COLD", "1") != "0"
                for bo, bi, enc, _, _ in cands:
                    d.cfg[5], d.cfg[6], d.cfg[7] = bo, bi, enc
                    launch(xs.data_ptr(), dys.data_ptr(), slab.data_ptr(), grad.data_ptr(), sp)
                    if cold:                  # dy comes from the preceding backward kernel (warm); x was written in the forward pass
                        t = self._timed_cold(lambda: launch(xs.data_ptr(), dys.data_ptr(), slab.data_ptr(), grad.data_ptr(), sp), [dys[:d.n * d.ho * d.wo * dy_stride * es]], Plan.tune_iters())
                    else:
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record(stream)
                        for _ in range(Plan.tune_iters()):
                            launch(xs.data_ptr(), dys.data_ptr(), slab.data_ptr(), grad.data_ptr(), sp)
                        b.record(stream)
                        b.synchronize()
                        t = a.elapsed_time(b)
                    if best is None or t < best[0]:
                        best = (t, (bo, bi, enc))
                hit = best[1]
            Plan._TUNE_CACHE[key] = hit
            Plan._tune_measured.add(key)
        d.cfg[5], d.cfg[6], d.cfg[7] = hit

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
            self._tune([dd], addend=kind, role="dgrad")
            # (networks with parallel branches: not gated -- their batch groups would not be, and a plan must compute the
            #  same sums whether its branches run as groups or on stream lanes)
            gi = self._gate_info.get(id(x)) if (self.bn_gate and first and len(descs) == 1 and self.n_lanes == 1 and self.es == 2) else None
            cfg = (C.c_int * 5)()
            if gi is not None and len(self._uses.get(id(x), [])) == 1 and x.c == x.c_valid and x.pixels * x.c * self.es <= self.bn_gate_bytes and \
                    self.lib.lh_igemm_config(C.byref(dd), self.dt, cfg) == 0 and (2 <= cfg[2] < 10 or 20 <= cfg[2] < 40):
                # x = relu(BN(raw)) with this convolution as its only consumer: the launch stores the ReLU-gated gradient
                # and the BatchNorm-backward partial sums of its tile (the node's backward skips its reduce pass)
                rows = self.lib.lh_igemm_stats_rows(C.byref(dd), self.dt)
                partial = self._alloc(rows * 2 * x.c, dtype=torch.float32)
                st = gi["st"]
                gate = _lib.BnBwdGate(gi["raw"].buf.data_ptr(), st["mean"].data_ptr(), st["invstd"].data_ptr(), st["scale"].data_ptr(),
                                      st["shift"].data_ptr(), partial.data_ptr())
                self.keep += [dd, gate]
                c = _Call(self.lib.lh_igemm_gated, (C.byref(dd), _ptr(dy), _ptr(pk), _ptr(dx), _ptr(addend), _ptr(amask), C.byref(gate), self.dt),
                          what + " + BN-backward gate")
                c.keep = dd
                c.ig = dict(src=1, dst=3, addend=4, addend_mask=5)
                self.bwd.append(c)
                self._gated[id(x)] = (partial, rows)
            else:
                self._igemm(self.bwd, dd, dy, pk, dx, addend, None, None, what, addend_mask=amask)
            yield dd, dd.ntaps

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
            elif 10 <= depth < 20:              # the wide-wave form of the 256 x 256 tile (four waves)
                wc, wp, depth = 2, 2, depth - 10
            if depth:
                return f"igemm_ring_kernel<{t}, {bm}, {bp}, {wc}, {wp}, {depth}, {kb}>"
            return f"igemm_kernel<{t}, {bm}, {bp}, {wc}, {wp}>"
        r = C.c_int(0)
        check(self.lib.lh_wgrad_tile(C.byref(d), wgrad[0], wgrad[1], self.dt, C.byref(a), C.byref(b), C.byref(c), C.byref(r)), "lh_wgrad_tile")
        wo, wi = {(256, 256): (2, 4), (128, 128): (2, 2), (128, 64): (4, 1), (64, 128): (1, 4), (64, 64): (2, 2)}[(a.value, b.value)]
        if r.value:
            return f"wgrad_ring_kernel<{t}, {a.value}, {b.value}, {wo}, {wi}, {r.value % 10}, {r.value // 10}>"
        return f"wgrad_kernel<{t}, {a.value}, {b.value}, {wo}, {wi}>"

    def _pending(self):
        return self._pend.setdefault(self._cur_lane, dict(calls=[], names=[], layers=0, ws=[], bytes=0))

    def _ws_note(self, setter, nbytes):
        ent = [setter, nbytes, self._cur_lane]
        self._ws_users.append(ent)
        if self.wgrad_group > 0:
            self._pending()["ws"].append(ent)
        return ent

    def _wl(self):
        """List that takes the weight-gradient work of the node being emitted (deferred group or the backward list)."""
        return self._pending()["calls"] if self.wgrad_group > 0 else self.bwd

    def _flush_wgrads(self, src, spread=False):
        """Append the deferred weight-gradient group of source lane `src` behind a 'wfork' event of that lane.
        spread: the LAST group of the backward pass -- nothing is left on the main stream to overlap it with, so its
        launches (each far from filling the machine) are dealt over all weight-gradient streams instead of queueing on one."""
        p = self._pend.get(src)
        if not p or not p["calls"]:
            return
        if self.wgrad_table:
            p["calls"] = self._table_wgrads(p["calls"])
        if self.wgrad_batch:
            p["calls"] = self._batch_wgrads(p["calls"])
        lanes = [-1 - ((self._w_flushes + i) % self._w_lanes) for i in range(self._w_lanes if spread else 1)]
        self._w_flushes += 1
        # units that must stay together on one stream, in order: a weight-gradient call with the small calls that follow
        # it (crop / unstage / bias), and the calls tagged to merge into one multi-problem launch
        fused, table_run, clusters = self.lib.lh_wgrad_fused, self.lib.lh_wgrad_table_run, []
        for c in p["calls"]:
            head = isinstance(c, _Call) and (c.fn is fused or c.fn is table_run)
            same = head and clusters and c.mtag is not None and getattr(clusters[-1][0], "mtag", None) is not None \
                and clusters[-1][0].mtag[:2] == c.mtag[:2]
            if clusters and (same or not head):
                clusters[-1].append(c)
            else:
                clusters.append([c])
        cost = {id(call): nbytes for _, call, _, _, nbytes in self.profile_meta}
        load = {L: 0.0 for L in lanes}
        where = {}
        for i in sorted(range(len(clusters)), key=lambda i: -sum(cost.get(id(c), 0.0) for c in clusters[i])):
            L = min(lanes, key=lambda L: (load[L], lanes.index(L)))
            where[i] = L
            load[L] += sum(cost.get(id(c), 0.0) for c in clusters[i]) + 1.0
        for L in lanes:
            mine = [clusters[i] for i in range(len(clusters)) if where[i] == L]
            if not mine:
                continue
            m = _Marker("wfork")
            m.slane, m.lane = L, src         # .lane of a wfork marker = the stream whose progress the group waits for
            self.bwd.append(m)
            for cl in mine:
                for c in cl:
                    c.slane = L
                    if getattr(c, "ws_ent", None) is not None:
                        c.ws_ent[2] = L      # the slab workspace follows the call's stream
                self.bwd += cl
        if p["names"]:
            self.bwd_marks.append((len(self.bwd), p["names"]))
        self._pend[src] = dict(calls=[], names=[], layers=0, ws=[], bytes=0)

    # kernel configurations (tile o, tile i, pixel rows per stage, ring depth) offered to a table of a tile class, best guess first
    # (measured, R50 bs 64: the 8-wave 256 x 256 tile with 64-row stages wins the >= 256-channel table by 20 % over 128 x 128; the
    #  layers with a side below 128 channels stream their operands -- 64 x 64 tiles are as fast for them as 128 x 64 / 64 x 128, and
    #  ONE class for all of them is one launch instead of three)
    _TABLE_CFGS = {
        (256, 256): ((256, 256, 64, 2), (256, 256, 32, 3), (128, 128, 64, 2)),
        (128, 128): ((128, 128, 64, 3), (128, 128, 64, 2), (128, 128, 32, 4)),
        (64, 64): ((64, 64, 64, 3), (64, 64, 64, 2), (64, 64, 32, 4)),
    }

    def _table_wgrads(self, calls):
        """The weight gradients of a deferred group are independent of each other and of everything else on their side stream: all of
        them that share a tile class become ONE lh_wgrad_table_run call -- one grid of the LDS-DMA weight-gradient kernel over a device
        table of argument blocks, every layer with its own pixel-split count, plus at most one fold grid.  The deep-K layers of stages
        3-4 and the head then run split-free or nearly so (their tiles fill the machine together), and a stage costs two launches instead
        of two per layer.  Kernel configuration and work-item length are measured on the real operands (_tune_table).  Calls that do not
        fit (the stem's row fold, fp32) stay as they are; the small calls that follow a tabled gradient (crop, bias) follow its table."""
        fused = self.lib.lh_wgrad_fused
        units = []
        for c in calls:
            if isinstance(c, _Call) and c.fn is fused:
                units.append([c])
            elif units:
                units[-1].append(c)
            else:
                units.append([c])
        big = os.environ.get("LH_WGRAD_TABLE_BIG", "1") != "0"

        def cls(u):
            c = u[0]
            if not (isinstance(c, _Call) and c.fn is fused and c.wargs is not None and c.wargs[1] <= 1):
                return None
            n_out, n_in = c.wargs[5], c.wargs[6]
            if n_out % 8 or n_in % 8:
                return None
            if big and n_out >= 256 and n_in >= 256:
                return (256, 256)
            return (128, 128) if n_out >= 128 and n_in >= 128 else (64, 64)
        groups = {}
        for u in units:
            groups.setdefault(cls(u), []).append(u)
        # a layer that is alone in its class joins the group's table of the nearest class (a smaller tile first: it only costs the larger
        # layer some operand re-reads; a larger tile multiplies padding for the small layer, which streams its operands anyway) -- one
        # launch + fold less per straggler (R50: the head's 1x1, the projection of stage 2)
        order = [(256, 256), (128, 128), (64, 64)]
        for k in order if os.environ.get("LH_WGRAD_TABLE_STRAGGLERS", "1") != "0" else ():
            if k in groups and len(groups[k]) == 1:
                i = order.index(k)
                hosts = [h for h in order[i + 1:] + order[:i][::-1] if h in groups and len(groups[h]) >= 2]
                if hosts:
                    groups[hosts[0]] += groups.pop(k)
        out, rest = [], []
        for k, us in groups.items():
            if k is None or len(us) < 2:
                rest += us
                continue
            out.append(self._make_table(k, us))
            for u in us:
                out += u[1:]
        for u in units:                      # the others keep their order
            if any(u is r for r in rest):
                out += u
        return out

    def _make_table(self, tile_class, units):
        lib = self.lib
        members = [u[0] for u in units]
        n = len(members)
        arr = (_lib.WgradCall * n)()
        for i, c in enumerate(members):
            a = c.wargs
            arr[i].d, arr[i].rows, arr[i].x, arr[i].dy, arr[i].dy_pix_stride, arr[i].n_out, arr[i].n_in = C.pointer(a[0]._obj), *a[1:7]
            arr[i].workspace = None
            arr[i].grad, arr[i].so, arr[i].si, arr[i].sr, arr[i].ss = a[8:13]
            arr[i].taps_rs, arr[i].accumulate = C.cast(a[13], C.POINTER(C.c_int)), a[14]
        cands = [cf for cf in Plan._TABLE_CFGS[tile_class]]
        cfg, target = self._tune_table(arr, members, cands)
        info, blob, ws = self._build_table(arr, n, cfg, target)
        names = [c.what.replace(" wgrad", "") for c in members]
        call = _Call(lib.lh_wgrad_table_run, (blob.data_ptr(), C.byref(info), self.dt), f"{n} x wgrad (table)", keep=(arr, info, blob, ws, members), lane=1)
        # bookkeeping: the members leave the shared-slab users and the profile attribution; the table takes their sums
        gone = {id(c.ws_ent) for c in members if c.ws_ent is not None}
        self._ws_users = [e for e in self._ws_users if id(e) not in gone]
        for pend in self._pend.values():
            pend["ws"] = [e for e in pend["ws"] if id(e) not in gone]
        ids = {id(c) for c in members}
        ms = [m for m in self.profile_meta if id(m[1]) in ids]
        self.profile_meta = [m for m in self.profile_meta if id(m[1]) not in ids]
        t = {"fp32": "float", "bf16": "__bf16", "fp16": "_Float16"}[self.precision]
        wo, wi = {(256, 256): (2, 4), (128, 128): (2, 2), (128, 64): (4, 1), (64, 128): (1, 4), (64, 64): (2, 2)}[(info.bo, info.bi)]
        self.profile_meta.append(("bwd", call, f"wgrad_ring_table_kernel<{t}, {info.bo}, {info.bi}, {wo}, {wi}, {info.depth}, {info.kps}>",
                                  sum(m[3] for m in ms), sum(m[4] for m in ms)))
        self.wgrad_tables.append((call, info, names))
        return call

    def _build_table(self, arr, n, cfg, target):
        """(info, device blob, slab workspace) of one table: size query, allocation, build on the host, upload."""
        lib = self.lib
        info = _lib.WgradTableInfo()
        cfg4 = (C.c_int * 4)(*cfg)
        check(lib.lh_wgrad_table_build(arr, n, self.dt, cfg4, target, None, None, 0, C.byref(info)), "lh_wgrad_table_build (sizes)")
        ws = torch.empty(info.workspace_bytes, dtype=torch.uint8, device=self.device)
        host = (C.c_ubyte * info.table_bytes)()
        check(lib.lh_wgrad_table_build(arr, n, self.dt, cfg4, target, ws.data_ptr(), host, info.table_bytes, C.byref(info)), "lh_wgrad_table_build")
        blob = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(self.device)
        return info, blob, ws

    def _tune_table(self, arr, members, cands):
        """Measured (kernel configuration, work-item length in ring stages) of one table: every offered configuration x a ladder of item
        lengths (split-free, 1/2, 1/3 ... of the longest member's stage count, and the library's automatic choice), timed on the
        members' REAL operand buffers filled with random bits for the measurement (cold caches: a deferred group runs long after its
        operands were written).  The choice fixes every member's split count, i.e. the fp32 summation order: cached per table signature."""
        n = len(members)
        if os.environ.get("LH_AUTOTUNE", "1") == "0":
            return cands[0], 0
        forced = os.environ.get("LH_WGRAD_TABLE_FORCE")       # experiments: "bo,bi,kps,depth,target"
        if forced:
            v = [int(t) for t in forced.split(",")]
            return tuple(v[:4]), v[4]
        key = ("wt", self.dt, tuple((self._desc_key(c.wargs[0]._obj), c.wargs[4], c.wargs[5], c.wargs[6]) for c in members))
        hit = Plan._TUNE_CACHE.get(key)
        if hit is not None and tuple(hit[:4]) in cands:
            return tuple(hit[:4]), hit[4]
        stream = torch.cuda.current_stream()
        sp = stream.cuda_stream
        bufs, saved = {}, []
        for c in members:
            for t in c.wbufs:
                bufs[t.data_ptr()] = t
        for t in bufs.values():
            saved.append((t, t.clone()))
            t.view(torch.int16).random_(-16000, 16000)
        best = None
        try:
            for cfg in cands:
                kps = cfg[2]
                smax = max((c.wargs[0]._obj.n * c.wargs[0]._obj.ho * c.wargs[0]._obj.wo + kps - 1) // kps for c in members)
                ladder = [0] + sorted({max(256 // kps, -(-smax // q)) for q in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64)}, reverse=True)
                seen = set()
                for target in ladder:
                    info, blob, ws = self._build_table(arr, n, cfg, target)
                    sig = (info.n_items, info.n_fold_items, info.workspace_bytes)
                    if sig in seen:
                        continue
                    seen.add(sig)
                    run = lambda: check(self.lib.lh_wgrad_table_run(blob.data_ptr(), C.byref(info), self.dt, sp), "autotune lh_wgrad_table_run")
                    run()
                    t = self._timed_cold(run, [], Plan.tune_iters())
                    if os.environ.get("LH_WGRAD_TABLE_LOG"):
                        print(f"[table {n} x wgrad] cfg {cfg} target {target:5d} items {info.n_items:5d} fold {info.n_fold_items:5d} "
                              f"slab {info.workspace_bytes >> 20:4d} MiB nsplit<= {info.nsplit_max:3d}: {t / Plan.tune_iters() * 1e3:8.1f} us", flush=True)
                    if best is None or t < best[0]:
                        best = (t, cfg, info.target_stages if target else 0)
                    del blob, ws
        finally:
            for t, keep in saved:
                t.copy_(keep)
        hit = tuple(best[1]) + (best[2],)
        Plan._TUNE_CACHE[key] = hit
        Plan._tune_measured.add(key)
        return tuple(hit[:4]), hit[4]

    def _batch_wgrads(self, calls):
        """The weight gradients of a deferred group are independent of each other and of everything else on their side
        stream: launches of the SAME shape and kernel plan (the repeated blocks of a stage) are brought next to each other
        and tagged to merge four at a time into lh_wgrad_fused_multi -- one weight-gradient launch and one fold launch for
        four layers.  The small late-stage layers (a few hundred workgroups, 20-40 us each) fill the machine together."""
        fused = self.lib.lh_wgrad_fused
        units, keys = [], []
        for c in calls:
            if isinstance(c, _Call) and c.fn is fused and c.mtag is None:
                units.append([c])
            elif units and not (isinstance(c, _Call) and c.fn is fused):
                units[-1].append(c)
            else:
                units.append([c])
        def key(u):
            c = u[0]
            if not (isinstance(c, _Call) and c.fn is fused and c.mtag is None and len(u) == 1 and c.keep_desc is not None):
                return None
            d = c.keep_desc
            return (self._desc_key(d), d.cfg[5], d.cfg[6], d.cfg[7])
        order, out = {}, []
        for u in units:
            k = key(u)
            order.setdefault(k if k is not None else ("single", id(u)), []).append(u)
        for k, us in order.items():
            if isinstance(k, tuple) and k and k[0] == "single" or len(us) < 2:
                for u in us:
                    out += u
                continue
            for i0 in range(0, len(us), 4):
                chunk = us[i0:i0 + 4]
                if len(chunk) >= 2:
                    self._n_groups += 1
                    for j, u in enumerate(chunk):
                        u[0].mtag = (("wb", self._n_groups), "w", j, 0)
                for u in chunk:
                    out += u
        return out

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
                    self._cur_lane = lane
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

    def _batch_order(self):
        """Compile order of the nodes: a list of items, each a list of node indices.  A fork .. join region (chains that are
        independent of each other, one per lane) is re-ordered position by position: the heads of the chains that are nodes
        of the same kind (convolution / BN-ReLU node) form ONE item, a batch group; every chain keeps its own order, so
        every dependency holds.  The region then runs on the main stream (its markers become no-ops)."""
        items, i, n = [], 0, len(self.nodes)
        while i < n:
            if self.nodes[i][0] != "fork" or not self.batch:
                items.append([i])
                i += 1
                continue
            j = i + 1
            while self.nodes[j][0] != "join":
                j += 1
            lanes = {}
            for t in range(i + 1, j):
                lanes.setdefault(self.node_lanes[t], []).append(t)
            if len(lanes) < 2:
                items += [[t] for t in range(i, j + 1)]
                i = j + 1
                continue
            # LH_BATCH=2: two half-groups (branches 0-1 | the rest) on two stream lanes, each merged pairwise -- the deep-K,
            # few-workgroup convolutions of the low-resolution branches then overlap the wide shallow ones of the others
            order = sorted(lanes)
            halves = [order[:2], order[2:]] if self.batch_split and len(order) >= 3 else [order]
            if len(halves) == 1:
                self.nodes[i] = self.nodes[j] = ("nop", {})
            items.append([i])
            for hi, half in enumerate(halves):
                for L in half:
                    for t in lanes[L]:
                        self.node_lanes[t] = hi
                queues = [list(lanes[L]) for L in half]
                while any(queues):
                    heads = {}
                    for q in queues:
                        if q:
                            heads.setdefault(self.nodes[q[0]][0], []).append(q)
                    kind = max(heads, key=lambda k: (len(heads[k]), k == "conv"))
                    qs = heads[kind]
                    if len(qs) >= 2 and kind in ("conv", "fuse"):
                        items.append([q.pop(0) for q in qs])
                    else:
                        items.append([qs[0].pop(0)])
            items.append([j])
            i = j + 1
        return items

    # ---- batch groups: joint kernel choice, then merging of the members' launches ------------------------------------
    def _conv_descs(self, nd):
        """Forward and (stride 1) data-gradient descriptor of a convolution node, as _c_conv builds them."""
        x, y, k, s, p = nd["x"], nd["y"], nd["k"], nd["s"], nd["p"]
        wt = self.params[nd["w"] + ".weight"]
        cout, cin = wt.shape[0], wt.shape[1]
        all_rs = [(r, q) for r in range(k) for q in range(k)]
        fwd = _desc(x.n, x.h, x.w, x.c, cin, y.h, y.w, s, s, y.c, y.h, y.w, 1, 1, 0, 0, y.c, [(r - p, q - p) for r, q in all_rs])
        dg = None
        if s == 1 and x.needs_grad and self.with_bwd:
            dg = _desc(y.n, y.h, y.w, y.c, y.c if y.c != cout else cout, x.h, x.w, 1, 1, x.c, x.h, x.w, 1, 1, 0, 0, x.c,
                       [(p - r, p - q) for r, q in all_rs])
        return fwd, dg

    _MULTI_TILES = ((64, 64), (64, 128), (128, 64), (128, 128))       # tiles the multi-problem kernels are instantiated for

    def _tune_group(self, nds):
        """ONE kernel configuration for the launches of a batch group of convolutions that will merge (forward, data
        gradient, weight gradient): the merged launch needs a common tile, so the members are not tuned one by one --
        every configuration that fits all of them is timed on the merged launch (scratch operands, cold caches).
        Returns the forced choices _tune / _tune_wgrad pick up while the members compile."""
        if os.environ.get("LH_AUTOTUNE", "1") == "0" or Plan.force_cfg is not None or Plan.force_wgrad is not None:
            return None
        if any(nd["x"].is_image for nd in nds):
            return None
        descs = [self._conv_descs(nd) for nd in nds]
        forced = {"member": 0}
        es = self.es
        for role, idx in (("fwd", 0), ("dgrad", 1)):
            ds = [d[idx] for d in descs]
            if any(d is None for d in ds):
                continue
            with_stats = role == "fwd" and self.training and all(id(nd["y"]) in self._bn_inputs for nd in nds)
            key = ("g", role, self.dt, with_stats) + tuple(self._desc_key(d) for d in ds)
            common = None
            for d in ds:
                buf, n = self._igemm_candidates(d)
                c = {tuple(buf[5 * i:5 * i + 4]) for i in range(n)}
                common = c if common is None else common & c
            cands = sorted(c + (0,) for c in (common or ()) if 2 <= c[2] < 10 and (c[0], c[1]) in self._MULTI_TILES)      # 4-wave tiled forms (the multi-problem kernels)
            # MIXED launches (igemm_mixed_kernel.h; experiment, LH_MIXED=1): the members the direct 3x3 kernel takes (C = 32 / 64
            # per tap: HRNet's two high-resolution branches) run its body inside the merged grid, the others the 64 x 128 ring
            # tile -- whose stage size the 64-byte K run of the 32-channel member no longer dictates.  Candidate = (tile
            # configuration of the ring members, 1).  MEASURED (round 4, HRNet-W32 bs 32 fp16, tuned from scratch): the tuner
            # prefers the mixed form in 3 of 26 groups, step 13.30-13.35 vs 13.27 ms -- the direct body's 156 KB of LDS leave
            # one workgroup per CU for the whole grid (832 workgroups = 3.25 rounds); off by default.
            direct = []
            for d in ds:
                buf, n = self._igemm_candidates(d)
                direct.append(next((tuple(buf[5 * i:5 * i + 4]) for i in range(n) if buf[5 * i + 2] == 100), None))
            mixed = os.environ.get("LH_MIXED", "0")        # "1": every member the direct kernel takes; "32": only the 32-channel ones (79 KB of LDS: two workgroups per CU)
            if mixed == "32":
                direct = [dc if dc is not None and dc[3] == 32 else None for dc in direct]
            if mixed in ("1", "32") and any(direct) and len(ds) >= 2:
                rest = None
                for d, dc in zip(ds, direct):
                    if dc is None:
                        buf, n = self._igemm_candidates(d)
                        c = {tuple(buf[5 * i:5 * i + 4]) for i in range(n)}
                        rest = c if rest is None else rest & c
                ring = sorted(c for c in rest if (c[0], c[1]) == (64, 128) and 2 <= c[2] < 10) if rest is not None else [(64, 128, 2, 64)]
                cands += [c + (1,) for c in ring]
            hit = Plan._TUNE_CACHE.get(key)
            if hit is not None and len(hit) == 4:
                hit = tuple(hit) + (0,)               # entries of earlier rounds: one tiled configuration for all members
            if hit is not None and hit not in cands:
                hit = None

            def per_member(cfg):
                return [dc if (cfg[4] and dc is not None) else cfg[:4] for dc in direct]
            if hit is None and cands:
                arr = (_lib.IgemmCall * len(ds))()
                warm = []
                for i, d in enumerate(ds):
                    kpad = (d.k_run * es + 127) // 128 * 128
                    src = self._scratch(f"g{i}in", d.n * d.hi * d.wi * d.in_pix_stride * es + 256)
                    arr[i].d = C.pointer(d)
                    arr[i].in_ = src.data_ptr()
                    arr[i].wpack = self._scratch(f"g{i}pack", (d.cout + 255) // 256 * 256 * max(d.ntaps, 1) * kpad + 256).data_ptr()
                    arr[i].out = self._scratch(f"g{i}out", d.n * d.OH * d.OW * d.out_pix_stride * es + 256).data_ptr()
                    if with_stats:
                        arr[i].stats = self._scratch(f"g{i}stats", max((d.n * d.ho * d.wo + 63) // 64, 1024) * 2 * d.cout * 4 + 256).data_ptr()
                    warm.append(src[:d.n * d.hi * d.wi * d.in_pix_stride * es])
                sp = torch.cuda.current_stream().cuda_stream

                def run():
                    check(self.lib.lh_igemm_multi(arr, len(ds), self.dt, sp), "group autotune lh_igemm_multi")
                best = None
                for cfg in cands:
                    for d, mc in zip(ds, per_member(cfg)):
                        d.cfg[0], d.cfg[1], d.cfg[2], d.cfg[3] = mc
                    run()
                    t = self._timed_cold(run, warm, Plan.tune_iters())
                    if os.environ.get("LH_TUNE_LOG"):
                        print(f"[tune {role or ''} {lead.k_run}x{lead.ntaps}->{lead.cout} M={lead.n * lead.ho * lead.wo} addend={addend}] cfg {cfg}: "
                              f"{t / Plan.tune_iters() * 1e3:7.1f} us", flush=True)
                    if best is None or t < best[0]:
                        best = (t, cfg)
                hit = best[1]
                Plan._TUNE_CACHE[key] = hit
                Plan._tune_measured.add(key)
            if hit is not None:
                forced[role] = per_member(hit) if hit[4] else hit[:4]
        # weight gradient: common (tile, stage rows, ring depth); per member the plan with the fewest workgroups -- the batch
        # fills the machine, a member need not
        if self.with_bwd:
            per, common = [], None
            for (d, _), nd in zip(descs, nds):
                y, wt = nd["y"], self.params[nd["w"] + ".weight"]
                buf = (C.c_int * (5 * 320))()
                n = self.lib.lh_wgrad_candidates(C.byref(d), y.c, wt.shape[1], self.dt, buf, 320)
                cs = [tuple(buf[5 * i:5 * i + 5]) for i in range(n)]
                per.append(cs)
                keys = {(c[0], c[1], (c[2] >> 16) & 255, (c[2] >> 24) & 255) for c in cs if c[0] <= 128 and c[1] <= 128}
                common = keys if common is None else common & keys
            key = ("gw", self.dt) + tuple(self._desc_key(d) for d, _ in descs)
            hit = Plan._TUNE_CACHE.get(key)
            if hit is not None and (len(hit) != 5 or tuple(hit[:4]) not in (common or ()) or len(hit[4]) != len(per)
                                    or any((hit[0], hit[1], enc) not in {c[:3] for c in cs} for enc, cs in zip(hit[4], per))):
                hit = None                                    # stale entry (tile or a member's split encoding no longer offered): measure again
            if hit is None and common:
                arr = (_lib.WgradCall * len(nds))()
                warm, keep = [], []
                for i, ((d, _), nd) in enumerate(zip(descs, nds)):
                    y, wt, k = nd["y"], self.params[nd["w"] + ".weight"], nd["k"]
                    cin = wt.shape[1]
                    rs = _taps_array([(r, q) for r in range(k) for q in range(k)])
                    dys = self._scratch(f"g{i}dy", d.n * d.ho * d.wo * y.c * es + 256)
                    slab = self._scratch(f"g{i}wws", (max(c[4] for c in per[i]) + 1) << 20)
                    arr[i].d, arr[i].rows = C.pointer(d), 0
                    arr[i].x = self._scratch(f"g{i}in", d.n * d.hi * d.wi * d.in_pix_stride * es + 256).data_ptr()
                    arr[i].dy, arr[i].dy_pix_stride, arr[i].n_out, arr[i].n_in = dys.data_ptr(), y.c, y.c, cin
                    arr[i].workspace = slab.data_ptr()
                    arr[i].grad = self._scratch(f"g{i}stats", y.c * cin * k * k * 4 + 256).data_ptr()
                    arr[i].so, arr[i].si, arr[i].sr, arr[i].ss = cin * k * k, k * k, k, 1
                    arr[i].taps_rs = C.cast(rs, C.POINTER(C.c_int))
                    keep.append(rs)
                    warm.append(dys[:d.n * d.ho * d.wo * y.c * es])
                sp = torch.cuda.current_stream().cuda_stream

                def runw():
                    check(self.lib.lh_wgrad_fused_multi(arr, len(nds), self.dt, sp), "group autotune lh_wgrad_fused_multi")
                best = None
                for tk in sorted(common):
                    for policy in (0, 1):                 # fewest workgroups per member / next larger split count
                        encs = []
                        for cs in per:
                            opts = sorted((c for c in cs if (c[0], c[1], (c[2] >> 16) & 255, (c[2] >> 24) & 255) == tk), key=lambda c: c[3])
                            encs.append(opts[min(policy, len(opts) - 1)][2])
                        for (d, _), enc in zip(descs, encs):
                            d.cfg[5], d.cfg[6], d.cfg[7] = tk[0], tk[1], enc
                        runw()
                        t = self._timed_cold(runw, warm, Plan.tune_iters())
                        if best is None or t < best[0]:
                            best = (t, tk + (tuple(encs),))
                hit = best[1]
                Plan._TUNE_CACHE[key] = hit
                Plan._tune_measured.add(key)
            if hit is not None:
                forced["wgrad"] = [(hit[0], hit[1], enc) for enc in hit[4]]
        return forced

    _MULTI_FN = None

    def _merge_groups(self):
        """Final pass of _compile: inside every run of launches that belong to one batch group, the k-th launch of each
        member merges with the others' into one lh_*_multi call when they are the same C-ABI function (and, for the
        convolutions, resolve to the same kernel configuration).  Members are independent of each other, so ordering the
        run position by position is legal.  Weight-gradient slabs and BN-backward workspaces, shared one after another on a
        stream by single launches, are handed out side by side to the members of a merged call."""
        lib = self.lib
        mergeable = (lib.lh_igemm, lib.lh_bn_finalize, lib.lh_fuse_fwd, lib.lh_fuse_bwd, lib.lh_wgrad_fused)     # ctypes functions do not hash
        pools, binds = {}, []               # (kind, stream lane) -> bytes needed; (struct array, index, field, pool key, offset)
        meta = {id(c): (name, fl, nb) for _, c, name, fl, nb in self.profile_meta}

        def build(fn, calls):
            n = len(calls)
            what = calls[0].what.split(" ")[-1] if fn is not lib.lh_igemm else " ".join(calls[0].what.split(" ")[1:])
            what = f"{n} x {what}"
            if fn is lib.lh_igemm:
                cfgs = set()
                for c in calls:
                    cfg = (C.c_int * 5)()
                    check(lib.lh_igemm_config(c.args[0], self.dt, cfg), "lh_igemm_config")
                    cfgs.add(tuple(cfg[:4]))
                ring = {c for c in cfgs if c[2] != 100}
                if len(ring) > 1 or any(not 2 <= c[2] < 10 or (c[0], c[1]) not in self._MULTI_TILES for c in ring):
                    return None
                if len(ring) != len(cfgs) and (os.environ.get("LH_MIXED", "0") not in ("1", "32") or any((c[0], c[1]) != (64, 128) for c in ring)):
                    # direct 3x3 members share a launch with the 64 x 128 tile only (igemm_mixed_kernel.h), and only when the
                    # mixed launch was asked for: members tuned one by one may pick the direct kernel in a default build, where
                    # the mixed kernel (measured slower) must not run -- they are launched one by one instead
                    return None
                arr = (_lib.IgemmCall * n)()
                for i, c in enumerate(calls):
                    a = c.args
                    arr[i].d = C.pointer(a[0]._obj)
                    (arr[i].in_, arr[i].wpack, arr[i].out, arr[i].addend, arr[i].addend_mask, arr[i].bias, arr[i].scale, arr[i].shift,
                     arr[i].stats) = a[1:10]
                m = _Call(lib.lh_igemm_multi, (arr, n, self.dt), what, keep=[c.keep for c in calls])
            elif fn is lib.lh_bn_finalize:
                arr = (_lib.BnFinalizeCall * n)(*[_lib.BnFinalizeCall(*c.args) for c in calls])
                m = _Call(lib.lh_bn_finalize_multi, (arr, n), what)
            elif fn is lib.lh_fuse_fwd:
                arr = (_lib.FuseFwdCall * n)()
                for i, c in enumerate(calls):
                    a = c.args
                    arr[i].d, arr[i].out, arr[i].n, arr[i].h, arr[i].w, arr[i].c = C.pointer(a[0]._obj), a[1], a[2], a[3], a[4], a[5]
                m = _Call(lib.lh_fuse_fwd_multi, (arr, n, self.dt), what)
            elif fn is lib.lh_fuse_bwd:
                arr = (_lib.FuseBwdCall * n)()
                off = 0
                for i, c in enumerate(calls):
                    a = c.args
                    arr[i].d, arr[i].n, arr[i].h, arr[i].w, arr[i].c = C.pointer(a[0]._obj), a[1], a[2], a[3], a[4]
                    binds.append((arr, i, "workspace", ("f", calls[0].slane), off))
                    off += (lib.lh_fuse_bwd_workspace_bytes(a[1], a[2], a[3], a[4]) + 255) // 256 * 256
                pools[("f", calls[0].slane)] = max(pools.get(("f", calls[0].slane), 0), off)
                m = _Call(lib.lh_fuse_bwd_multi, (arr, n, self.dt), what)
            else:
                arr = (_lib.WgradCall * n)()
                off = 0
                for i, c in enumerate(calls):
                    a = c.args
                    arr[i].d, arr[i].rows, arr[i].x, arr[i].dy, arr[i].dy_pix_stride, arr[i].n_out, arr[i].n_in = C.pointer(a[0]._obj), *a[1:7]
                    arr[i].grad, arr[i].so, arr[i].si, arr[i].sr, arr[i].ss = a[8:13]
                    arr[i].taps_rs, arr[i].accumulate = C.cast(a[13], C.POINTER(C.c_int)), a[14]
                    binds.append((arr, i, "workspace", ("w", calls[0].slane), off))
                    off += (lib.lh_wgrad_workspace_bytes(a[0], a[5], a[6], self.dt) + 255) // 256 * 256
                pools[("w", calls[0].slane)] = max(pools.get(("w", calls[0].slane), 0), off)
                m = _Call(lib.lh_wgrad_fused_multi, (arr, n, self.dt), what, keep=[c.keep for c in calls], lane=calls[0].lane)
            m.slane = calls[0].slane
            self.keep += [arr] + list(calls)
            ms = [meta[id(c)] for c in calls if id(c) in meta]
            if ms:
                name = ms[0][0].replace("igemm_ring_kernel", "igemm_ring_multi_kernel").replace("wgrad_ring_kernel", "wgrad_ring_multi_kernel")
                self.profile_meta.append(("fwd" if calls[0].mtag[1] == "f" else "bwd", m, name, sum(x[1] for x in ms), sum(x[2] for x in ms)))
            return m

        def merged(lst):
            out, remap, i = [], {}, 0
            while i < len(lst):
                c = lst[i]
                tag = getattr(c, "mtag", None)
                if tag is None:
                    remap[i] = len(out)
                    out.append(c)
                    i += 1
                    continue
                j = i
                while j < len(lst) and getattr(lst[j], "mtag", None) is not None and lst[j].mtag[:2] == tag[:2]:
                    j += 1
                members = {}
                for c2 in lst[i:j]:
                    members.setdefault(c2.mtag[2], []).append(c2)
                chains = [members[k] for k in sorted(members)]
                new = []
                for k in range(max(len(ch) for ch in chains)):
                    row = [ch[k] for ch in chains if k < len(ch)]
                    m = None
                    if len(row) >= 2 and all(r.fn is row[0].fn for r in row) and any(row[0].fn is f for f in mergeable) and all(r.slane == row[0].slane for r in row):
                        m = build(row[0].fn, row)
                    new += [m] if m is not None else row
                for t in range(i, j):
                    remap[t] = len(out) + len(new)        # a position inside the run maps to the run's end
                out += new
                i = j
            remap[len(lst)] = len(out)
            return out, remap

        self.unmerged = (list(self.fwd), list(self.bwd))      # the same launches one by one (tests: bit-equal results)
        self.fwd, _ = merged(self.fwd)
        self.bwd, remap = merged(self.bwd)
        self.bwd_marks = [(remap[e], names) for e, names in self.bwd_marks]
        bufs = {k: self._alloc(max(v, 256), dtype=torch.uint8) for k, v in pools.items()}
        for arr, i, field, k, off in binds:
            setattr(arr[i], field, bufs[k].data_ptr() + off)

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

    def _fuse_head(self, nd, pack, bias):
        """Inference plans: `final_layer(relu(bn(deconv(x))))` (pose_resnet.py:245-246) as ONE launch.  When this 1x1
        convolution produces the network output and its only input is the output of a transposed convolution whose
        BatchNorm + ReLU were folded into its epilogue, the head is applied to every tile of that launch while it is in
        LDS (lh_igemm_phases_head): the C-channel activation is never written.  Returns False when the pattern does not
        apply (training plans, HRNet's head, more than 256 channels, a tile other than 256 x 256 on offer)."""
        x, y, k, s, p = nd["x"], nd["y"], nd["k"], nd["s"], nd["p"]
        if not Plan.fuse_head or self.with_bwd or self.es != 2 or k != 1 or s != 1 or p != 0 or y.c_valid > 32 or x.c > 256:
            return False
        if not any(kind == "output" and n["y"] is y for kind, n in self.nodes):
            return False
        users = sum(1 for kind, n in self.nodes
                    if (kind in ("conv", "deconv", "maxpool") and n["x"] is x) or (kind == "fuse" and any(a is x for a, _, _ in n["terms"])))
        prods = self._producers.get(id(x)) or []
        if users != 1 or len(prods) != 1 or prods[0].fn is not self.lib.lh_igemm_phases or prods[0] not in self.fwd:
            return False
        call = prods[0]
        a = call.args
        ig = self._IGP
        if a[ig["dst"]] != x.buf.data_ptr() or a[ig["addend"]] or a[ig["bias"]] or a[ig["stats"]] or not a[ig["scale"]]:
            return False
        descs = call.keep
        lead = max(descs, key=lambda dd: dd.ntaps)
        if (lead.cfg[0], lead.cfg[1]) != (256, 256):           # the head lives in the 256 x 256 tile's epilogue
            buf, n = self._igemm_candidates(lead)
            big = [tuple(buf[5 * i:5 * i + 4]) for i in range(n) if (buf[5 * i], buf[5 * i + 1]) == (256, 256)]
            if not big:
                return False
            best = min(big, key=lambda c: (c[3] != 128, c[2]))
            for dd in descs:
                dd.cfg[0], dd.cfg[1], dd.cfg[2], dd.cfg[3] = best
        if not all(dd.relu == lead.relu for dd in descs):
            return False
        self.out_nchw = self._alloc(y.n, y.c_valid, y.h, y.w, dtype=torch.float32)
        kstep = 128 // self.es
        head = _lib.Head(pack.data_ptr(), ((x.c + kstep - 1) // kstep * kstep) * self.es, _ptr(bias), self.out_nchw.data_ptr(), y.c_valid)
        fused = _Call(self.lib.lh_igemm_phases_head, (a[0], a[1], a[ig["src"]], a[3], a[ig["scale"]], a[ig["shift"]], C.byref(head), self.dt),
                      call.what.replace("fwd", "fwd + head"), keep=call.keep)
        fused.slane = call.slane
        self.keep += [head, call]
        self.fwd[self.fwd.index(call)] = fused
        self._head_fused = True
        return True

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
                    self.profile_meta.append(("bwd", self.bwd[-1], self._kname(dd), 2.0 * dd.n * dd.ho * dd.wo * cin * cout * ntaps,
                                              (x.pixels * x.c + y.pixels * y.c) * self.es if batched else
                                              (dd.n * dd.ho * dd.wo * x.c + y.pixels * y.c / (s * s)) * self.es))
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
                    self.profile_meta.append(("bwd", self.bwd[-1], self._kname(dg), flops, (x.pixels * x.c + y.pixels * y.c) * self.es))
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
            call = _Call(self.lib.lh_fuse_bwd, None, "fuse bwd")

            def set_ws(ptr):
                args[5] = ptr
                call.args = tuple(args)
            self._ws_users_fuse.append((set_ws, self._cur_lane))
            self.bwd.append(call)
            # algorithmic bytes of the BN / ReLU backward of this node: the reduce pass reads dout and every BN term's x, the
            # apply pass reads them again and writes one gradient per term that takes one (5 tensor passes per BN)
            n_bn = sum(1 for _, bn, _ in terms if bn is not None)
            n_dx = sum(1 for i in range(len(terms)) if bd.dx[i])
            passes = (2 * (1 + n_bn) if n_bn else 1) + n_dx
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

    def _fold_eval_bn(self, terms, bn_state, out, relu):
        """Inference plans: BatchNorm uses running statistics, so `relu(BN(conv) [+ residual | + BN(conv_ds)])` is folded
        into the producing convolution's epilogue (scale/shift on the fp32 accumulator, addend, ReLU) and the
        elementwise pass disappears.  Returns False when the pattern does not apply (e.g. HRNet's upsampled terms)."""
        if self.with_bwd or len(terms) > 2 or any(l for _, _, l in terms) or terms[0][1] is None:
            return False
        prods = [self._producers.get(id(a)) if bn else None for a, bn, _ in terms]
        if prods[0] is None or (len(terms) == 2 and terms[1][1] is not None and prods[1] is None):
            return False
        if any(self._in_closed_region(c) for pl in prods if pl for c in pl):
            return False            # HRNet exchange sums: the producer runs on a branch lane, the other term on another
        if len(terms) == 2 and self._ready.get(id(terms[1][0]), 0) > min(self.fwd.index(c) for c in prods[0]):
            return False            # the other term is produced AFTER the convolution that would have to add it
        # the eval-affine launches of this node were appended to self.fwd just above: they only depend on the
        # weights, so they move to the pack list (run when weights change, not per forward)
        n_aff = sum(1 for _, bn, _ in terms if bn is not None)
        self.packs += self.fwd[-n_aff:]
        del self.fwd[-n_aff:]
        obuf = out.buf.data_ptr()
        main, st0 = prods[0], bn_state[0]
        addend = 0
        if len(terms) == 2:
            res_act, res_bn, _ = terms[1]
            addend = res_act.buf.data_ptr()
            if res_bn is not None:                # projection shortcut: BN folded into ITS conv, written in place
                st1 = bn_state[1]
                for c in prods[1]:
                    self._patch(c, relu=0, scale=st1["scale"].data_ptr(), shift=st1["shift"].data_ptr())
                # the shortcut must be complete before the main conv adds it
                last_res = max(self.fwd.index(c) for c in prods[1])
                for c in main:
                    i = self.fwd.index(c)
                    if i < last_res:
                        self.fwd.insert(last_res, self.fwd.pop(i))
        for c in main:
            self._patch(c, relu=relu, dst=obuf, addend=addend, scale=st0["scale"].data_ptr(), shift=st0["shift"].data_ptr())
        self._producers.setdefault(id(out), []).extend(main)       # `out` is now written by these launches (see _fuse_head)
        if len(terms) == 2 and relu:
            self._fuse_bottleneck(terms[0][0], out)
        return True

    def _fuse_bottleneck(self, y3, out):
        """Inference plans: a stride-1 bottleneck of the first ResNet stage (pose_resnet.py:61-99: conv1 1x1 -> bn1 -> relu ->
        conv2 3x3 -> bn2 -> relu -> conv3 1x1 -> bn3, + residual, relu; 64 mid channels, 256 out) as ONE launch
        (lh_bottleneck_infer): only the block input and the residual are read and the block output written, the 64-channel
        intermediates stay in LDS -- 2.4 instead of 4.8 GB per identity block at configs[4].  Called when the block's last
        node has just been folded into conv3's epilogue (_fold_eval_bn); walks back conv3 <- conv2 <- conv1 and replaces the
        three launches when every link is what the kernel implements.  The projection shortcut of the stage's first block
        stays a launch of its own (its output is the residual)."""
        if not Plan.fuse_bottleneck or self.with_bwd or self.training or self.es != 2:
            return False
        conv_of = lambda act: next((n for kind, n in self.nodes if kind == "conv" and n["y"] is act), None)
        users = lambda act: sum(1 for kind, n in self.nodes
                                if (kind in ("conv", "deconv", "maxpool") and n["x"] is act) or (kind == "fuse" and any(a is act for a, _, _ in n["terms"]))
                                or (kind == "output" and n["y"] is act))
        n3 = conv_of(y3)
        if n3 is None or (n3["k"], n3["s"], n3["p"]) != (1, 1, 0) or n3["bias"]:
            return False
        a2 = n3["x"]                                   # relu(bn2(conv2)): written by conv2's launch since its node was folded
        p2 = self._producers.get(id(a2)) or []
        n2 = conv_of(next((t[0] for kind, n in self.nodes if kind == "fuse" and n["out"] is a2 for t in n["terms"]), None))
        if len(p2) != 1 or n2 is None or (n2["k"], n2["s"], n2["p"]) != (3, 1, 1) or n2["bias"] or users(a2) != 1:
            return False
        a1 = n2["x"]
        p1 = self._producers.get(id(a1)) or []
        n1 = conv_of(next((t[0] for kind, n in self.nodes if kind == "fuse" and n["out"] is a1 for t in n["terms"]), None))
        if len(p1) != 1 or n1 is None or (n1["k"], n1["s"], n1["p"]) != (1, 1, 0) or n1["bias"] or users(a1) != 1:
            return False
        p3 = self._producers.get(id(y3)) or []
        if len(p3) != 1:
            return False
        c1, c2, c3 = p1[0], p2[0], p3[0]
        lib, ig = self.lib, self._IG
        if any(c.fn is not lib.lh_igemm or c not in self.fwd or self._in_closed_region(c) for c in (c1, c2, c3)):
            return False
        x = n1["x"]
        d1, d2, d3 = c1.keep, c2.keep, c3.keep
        ok = (d1.cout == 64 and d2.cout == 64 and d3.cout == 256 and d2.k_run == 64 and d3.k_run == 64 and d1.k_run == x.c and x.c % 32 == 0
              and 64 <= x.c <= 1024 and d1.in_pix_stride == x.c and d2.in_pix_stride == 64 and d3.in_pix_stride == 64
              and d1.relu == 1 and d2.relu == 1 and d3.relu == 1 and (d1.ho, d1.wo) == (x.h, x.w) and (d3.ho, d3.wo) == (x.h, x.w))
        a1_, a2_, a3_ = c1.args, c2.args, c3.args
        ok = ok and all(a[ig["scale"]] and a[ig["shift"]] and not a[ig["bias"]] and not a[ig["stats"]] and not a[ig["addend_mask"]] for a in (a1_, a2_, a3_))
        ok = ok and not a1_[ig["addend"]] and not a2_[ig["addend"]] and a3_[ig["addend"]] and a1_[ig["src"]] == x.buf.data_ptr()
        ok = ok and a3_[ig["dst"]] not in (a1_[ig["src"]], a3_[ig["addend"]])
        # lh_bottleneck_infer writes a DENSE 256-channel output and reads a dense residual: a block whose output is a strided or
        # sliced view (another pixel stride, a placement inside a larger image) keeps its three launches
        ok = ok and d3.out_pix_stride == 256 and (d3.OH, d3.OW, d3.osh, d3.osw, d3.ooh, d3.oow) == (x.h, x.w, 1, 1, 0, 0)
        ok = ok and d1.out_pix_stride == 64 and d2.out_pix_stride == 64
        if not ok:
            return False
        bd = _lib.BottleneckDesc(x.n, x.h, x.w, x.c, 64, 256)
        fused = _Call(lib.lh_bottleneck_infer, (C.byref(bd), a1_[ig["src"]], a1_[ig["pack"]], a2_[ig["pack"]], a3_[ig["pack"]],
                                                a1_[ig["scale"]], a1_[ig["shift"]], a2_[ig["scale"]], a2_[ig["shift"]], a3_[ig["scale"]], a3_[ig["shift"]],
                                                a3_[ig["addend"]], a3_[ig["dst"]], self.dt), c1.what.replace("conv1 fwd", "bottleneck fwd (conv1 + conv2 + conv3 + residual)"),
                      keep=[bd, d1, d2, d3, c1, c2, c3])
        fused.slane = c3.slane
        self.fwd[self.fwd.index(c3)] = fused
        self.fwd.remove(c1)
        self.fwd.remove(c2)
        flops = sum(fl for _, c, _, fl, _ in self.profile_meta if c in (c1, c2, c3))
        self.profile_meta = [m for m in self.profile_meta if m[1] not in (c1, c2, c3)]
        self.profile_meta.append(("fwd", fused, "bottleneck_infer_kernel", flops, (x.pixels * x.c + 2 * out.pixels * out.c) * self.es))
        self._producers[id(out)] = [fused]
        self._n_fused_bottlenecks = getattr(self, "_n_fused_bottlenecks", 0) + 1
        return True

    def _fuse_stem_pool(self, nd):
        """Inference plans: `maxpool(relu(bn1(conv1(x))))` (pose_resnet.py:151-156 and the first lines of its forward) as ONE
        launch (lh_stem_pool: direct 7x7 / stride 2 convolution with the weights in registers, the eval-mode BatchNorm
        folded into its epilogue, the 3x3 / stride 2 maximum taken from the tile in LDS) -- the 64-channel convolution
        output, the largest activation of the network, is never written.  Applies when the pool's input is produced by the
        stem convolution alone (BatchNorm + ReLU already folded into it by _fold_eval_bn) and has no other reader."""
        x, y = nd["x"], nd["y"]
        if not Plan.fuse_stem or self.with_bwd or self.training or self.es != 2 or x.c != 64 or x.c_valid != 64:
            return False
        users = sum(1 for kind, n in self.nodes
                    if (kind in ("conv", "deconv", "maxpool") and n["x"] is x) or (kind == "fuse" and any(a is x for a, _, _ in n["terms"]))
                    or (kind == "output" and n["y"] is x))
        prods = self._producers.get(id(x)) or []
        if users != 1 or len(prods) != 1 or prods[0].fn is not self.lib.lh_igemm or not prods[0].what.endswith("stem fwd") or prods[0] not in self.fwd:
            return False
        call = prods[0]
        a, ig, d = call.args, self._IG, call.keep
        if (d.ntaps, d.k_run, d.sh, d.sw, d.cout, d.relu) != (7, 32, 2, 2, 64, 1) or a[ig["dst"]] != x.buf.data_ptr() or a[ig["addend"]] or a[ig["stats"]]:
            return False
        ybuf = self._act_buf(y)
        fused = _Call(self.lib.lh_stem_pool, (a[ig["src"]], d.n, d.hi, d.wi, a[ig["pack"]], a[ig["bias"]], a[ig["scale"]], a[ig["shift"]],
                                              ybuf.data_ptr(), d.ho, d.wo, 1, self.dt), "conv1 stem fwd + maxpool", keep=d)
        fused.slane = call.slane
        self.keep.append(call)
        self.fwd[self.fwd.index(call)] = fused
        self.profile_meta = [(w, fused if c is call else c, "stem_pool_kernel" if c is call else nm, fl,
                              (nb - x.pixels * x.c * self.es + y.pixels * y.c * self.es) if c is call else nb) for w, c, nm, fl, nb in self.profile_meta]
        self._producers[id(y)] = [fused]
        return True

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
            if gi is not None and len(self._uses.get(id(x), [])) == 1 and x.c == x.c_valid and nch & (nch - 1) == 0 and nch <= 256:
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
