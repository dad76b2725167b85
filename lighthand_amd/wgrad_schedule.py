"""Where the weight gradients of a plan run: deferred groups behind an event on side streams, table launches (one grid per tile class of
a group, a split count per layer: lh_wgrad_table_run) and same-shape merging of what is left.  A mixin of ``engine.Plan`` (split out of
engine.py in round 6).  Reference: loss.backward(), src/utils/method.py:182."""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import FuseBwdDesc, FuseDesc, IgemmDesc, check
from .graph import Act, _Call, _Marker, _desc, _ptr, _taps_array


class WgradSchedule:
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
        cands = [cf for cf in type(self)._TABLE_CFGS[tile_class]]
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
