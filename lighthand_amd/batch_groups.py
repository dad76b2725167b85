"""Multi-problem launches for networks with parallel branches (HighResolutionModule, src/modeling/hrnet/pose_hrnet.py:101-265): the
compile order that puts the k-th node of every branch side by side, and the final pass that merges their launches into lh_*_multi calls.
A mixin of ``engine.Plan`` (split out of engine.py in round 6)."""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import FuseBwdDesc, FuseDesc, IgemmDesc, check
from .graph import Act, _Call, _Marker, _desc, _ptr, _taps_array


class BatchGroups:
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
            # (two half-groups on two stream lanes -- LH_BATCH=2, round 2 -- were measured slower, 17.0 vs 16.0 ms, and removed in round 6)
            order = sorted(lanes)
            self.nodes[i] = self.nodes[j] = ("nop", {})
            items.append([i])
            for L in order:
                for t in lanes[L]:
                    self.node_lanes[t] = 0
            queues = [list(lanes[L]) for L in order]
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
                if len(ring) != len(cfgs):
                    # members tuned one by one may pick the direct 3x3 kernel: direct and tiled members do not share a launch (the mixed
                    # grid of round 4 was measured slower and removed) -- they are launched one by one instead
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
