"""Measured kernel choice (the plan's autotuner): every compiled-in configuration that fits a launch is timed on operands of the real
size in the cache state the launch meets inside a step; choices persist in a versioned cache file and a shipped database.  A mixin of
``engine.Plan`` (split out of engine.py in round 6): convolution launches (``_tune``), per-layer weight gradients (``_tune_wgrad``), table
launches of the weight gradient (``_tune_table``) and the joint choice of HRNet's batch groups (``_tune_group``)."""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import FuseBwdDesc, FuseDesc, IgemmDesc, check
from .graph import Act, _Call, _Marker, _desc, _ptr, _taps_array


def _record_time(key, choice, ms):
    """LH_TUNE_TIMES=<file>: every timed candidate as a line (key, choice, milliseconds for the timed launches) -- tools/ensemble_tune.py adds the
    times of several sessions (boxes of the pool differ in what they favour among near-ties) and picks the candidate that is fastest in the sum."""
    path = os.environ.get("LH_TUNE_TIMES")
    if path:
        with open(path, "a") as f:
            f.write(repr((key, tuple(choice), float(ms))) + "\n")


class Tuner:
    # ------------------------------------------------------------------ kernel autotuning
    _TUNE_CACHE = {}        # launch signature -> (bm, bp, depth, kb): shared by every plan of the process

    # test hooks: force_cfg(candidates) -> (bm, bp, depth, kb) | None and force_wgrad(candidates) -> (bo, bi, enc) | None
    # replace the measurement for the plans built while they are set (tests walk every compiled-in configuration)
    force_cfg = None

    force_wgrad = None

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
        fb = type(self)._flush_buf.get(dev)
        if fb is None:
            fb = type(self)._flush_buf[dev] = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
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
        buf = (C.c_int * (5 * type(self)._MAX_CANDS))()
        n = self.lib.lh_igemm_candidates(C.byref(desc), self.dt, buf, type(self)._MAX_CANDS)
        if not 0 <= n < type(self)._MAX_CANDS:
            raise _lib.LightHandError(f"lh_igemm_candidates returned {n} entries for a buffer of {type(self)._MAX_CANDS}")
        return buf, n

    def _tune(self, descs, with_stats=False, addend=None, role=None, gate=None):
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
        key = (self.dt, with_stats) + tuple(self._desc_key(d) for d in descs) + ((addend,) if addend else ()) + (("gate-" + gate[0],) if gate else ())
        hit = type(self)._TUNE_CACHE.get(key) if type(self).force_cfg is None else None
        buf, n = self._igemm_candidates(lead)
        cands = [tuple(buf[5 * i:5 * i + 4]) for i in range(n)]
        if len(descs) > 1:
            cands = [c for c in cands if c[2] not in (1, 100)]  # the persistent kernels take single launches only
        if len(descs) > 1 and self._phase_rows(descs) <= 0:
            cands = []                                          # phases that cannot be batched: keep the default
        if hit is not None and hit != (0, 0, 0, 0) and hit not in cands:
            hit = None                                          # stale entry (configuration no longer compiled in): measure again
        if hit is None:
            if type(self).force_cfg is not None:
                forced = type(self).force_cfg(cands) if cands else None
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
                run_plain, run_gated, penalty = run, None, 0.0
                if gate is not None and len(descs) == 1:
                    nout = lead.n * lead.OH * lead.OW * lead.out_pix_stride
                    gx = self._scratch("gate_x", nout * es + 256)
                    gvec = self._scratch("gate_vec", 4 * lead.cout * 4 + 256).view(torch.float32)
                    gvec[:4 * lead.cout] = 1.0
                    gpart = self._scratch("stats", rows * 2 * lead.cout * 4 + 256)
                    gbits = self._scratch("gate_bits", nout * es // 16 + 256) if gate[0] != "x" else None
                    c_ = lead.cout
                    gt = _lib.BnBwdGate(gx.data_ptr(), gvec.data_ptr(), gvec.data_ptr() + 4 * c_, gvec.data_ptr() + 8 * c_, gvec.data_ptr() + 12 * c_,
                                        gpart.data_ptr(), _ptr(gbits))
                    if gate[0] == "mask2":
                        gt.x2, gt.mean2, gt.invstd2 = self._scratch("gate_x2", nout * es + 256).data_ptr(), gvec.data_ptr(), gvec.data_ptr() + 4 * c_
                        gt.partial2 = self._scratch("gate_part2", rows * 2 * lead.cout * 4 + 256).data_ptr()

                    def run_gated():
                        check(self.lib.lh_igemm_gated(C.byref(lead), src.data_ptr(), packs[0].data_ptr(), dst.data_ptr(), _ptr(add), _ptr(amask),
                                                      C.byref(gt), self.dt, sp), "autotune lh_igemm_gated")
                    # ms per launch: the reduce pass lh_fuse_bwd keeps (per BatchNorm term: dout and that term's x)
                    penalty = (4.0 if gate[0] == "mask2" else 2.0) * gate[1] / 4.5e12 * 1e3 + 2e-3
                best = None
                cold = os.environ.get("LH_TUNE_COLD", "1") != "0"
                for cfg in cands:
                    for d in descs:
                        d.cfg[0], d.cfg[1], d.cfg[2], d.cfg[3] = cfg
                    gated = run_gated is not None and self._cfg_gateable(cfg, gate[0], gate[1])
                    run = run_gated if gated else run_plain
                    run()
                    if cold:                  # input = the previous kernel's output (warm), everything else cold
                        t = self._timed_cold(run, [src[:lead.n * lead.hi * lead.wi * lead.in_pix_stride * es]], type(self).tune_iters())
                    else:
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record(stream)
                        for _ in range(type(self).tune_iters()):
                            run()
                        b.record(stream)
                        b.synchronize()
                        t = a.elapsed_time(b)
                    if run_gated is not None and not gated:
                        t += penalty * type(self).tune_iters()
                    _record_time(key, cfg, t)
                    if os.environ.get("LH_TUNE_LOG"):
                        print(f"[tune {role or ''} {lead.k_run}x{lead.ntaps}->{lead.cout} M={lead.n * lead.ho * lead.wo} addend={addend} gate={gate and gate[0]}{'' if gate is None else ('+' if gated else '-')}] cfg {cfg}: "
                              f"{t / type(self).tune_iters() * 1e3:7.1f} us", flush=True)
                    if best is None or t < best[0]:
                        best = (t, cfg)
                hit = best[1]
            type(self)._TUNE_CACHE[key] = hit
            type(self)._tune_measured.add(key)
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
        hit = type(self)._TUNE_CACHE.get(key) if type(self).force_wgrad is None else None
        buf = (C.c_int * (5 * 320))()
        n = self.lib.lh_wgrad_candidates(C.byref(d), n_out, n_in, self.dt, buf, 320)
        cands = [tuple(buf[5 * i:5 * i + 5]) for i in range(n)]
        if hit is not None and hit != (0, 0, 0) and hit not in [c[:3] for c in cands]:
            hit = None
        if hit is None:
            if type(self).force_wgrad is not None:
                d.cfg[5], d.cfg[6], d.cfg[7] = (type(self).force_wgrad(cands) if cands else None) or (0, 0, 0)
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
                        t = self._timed_cold(lambda: launch(xs.data_ptr(), dys.data_ptr(), slab.data_ptr(), grad.data_ptr(), sp), [dys[:d.n * d.ho * d.wo * dy_stride * es]], type(self).tune_iters())
                    else:
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record(stream)
                        for _ in range(type(self).tune_iters()):
                            launch(xs.data_ptr(), dys.data_ptr(), slab.data_ptr(), grad.data_ptr(), sp)
                        b.record(stream)
                        b.synchronize()
                        t = a.elapsed_time(b)
                    _record_time(key, (bo, bi, enc), t)
                    if best is None or t < best[0]:
                        best = (t, (bo, bi, enc))
                hit = best[1]
            type(self)._TUNE_CACHE[key] = hit
            type(self)._tune_measured.add(key)
        d.cfg[5], d.cfg[6], d.cfg[7] = hit

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
        # a table whose whole work is a few tens of microseconds (test-sized models) is not worth a measurement: the library's automatic choice
        work = sum(-(-c.wargs[0]._obj.n * c.wargs[0]._obj.ho * c.wargs[0]._obj.wo // 64) * -(-c.wargs[5] // cands[0][0]) * -(-c.wargs[6] // cands[0][1])
                   * c.wargs[0]._obj.ntaps for c in members)
        if work < int(os.environ.get("LH_WGRAD_TABLE_TUNE_MIN", "20000")):
            return cands[0], 0
        key = ("wt", self.dt, tuple((self._desc_key(c.wargs[0]._obj), c.wargs[4], c.wargs[5], c.wargs[6]) for c in members))
        hit = type(self)._TUNE_CACHE.get(key)
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
                    t = self._timed_cold(run, [], type(self).tune_iters())
                    _record_time(key, tuple(cfg) + (info.target_stages if target else 0,), t)
                    if os.environ.get("LH_WGRAD_TABLE_LOG"):
                        print(f"[table {n} x wgrad] cfg {cfg} target {target:5d} items {info.n_items:5d} fold {info.n_fold_items:5d} "
                              f"slab {info.workspace_bytes >> 20:4d} MiB nsplit<= {info.nsplit_max:3d}: {t / type(self).tune_iters() * 1e3:8.1f} us", flush=True)
                    if best is None or t < best[0]:
                        best = (t, cfg, info.target_stages if target else 0)
                    del blob, ws
        finally:
            for t, keep in saved:
                t.copy_(keep)
        hit = tuple(best[1]) + (best[2],)
        type(self)._TUNE_CACHE[key] = hit
        type(self)._tune_measured.add(key)
        return tuple(hit[:4]), hit[4]

    def _tune_group(self, nds):
        """ONE kernel configuration for the launches of a batch group of convolutions that will merge (forward, data
        gradient, weight gradient): the merged launch needs a common tile, so the members are not tuned one by one --
        every configuration that fits all of them is timed on the merged launch (scratch operands, cold caches).
        Returns the forced choices _tune / _tune_wgrad pick up while the members compile."""
        if os.environ.get("LH_AUTOTUNE", "1") == "0" or type(self).force_cfg is not None or type(self).force_wgrad is not None:
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
            # (a 5th element of a candidate marked round 4's MIXED launch -- direct 3x3 bodies inside the merged grid; measured slower, removed in
            #  round 6: it is always 0 now and stays in the tuple so that the shipped database's entries keep their form)
            direct = [None] * len(ds)
            hit = type(self)._TUNE_CACHE.get(key)
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
                    t = self._timed_cold(run, warm, type(self).tune_iters())
                    _record_time(key, cfg, t)
                    if os.environ.get("LH_TUNE_LOG"):
                        print(f"[tune {role or ''} {lead.k_run}x{lead.ntaps}->{lead.cout} M={lead.n * lead.ho * lead.wo} addend={addend}] cfg {cfg}: "
                              f"{t / type(self).tune_iters() * 1e3:7.1f} us", flush=True)
                    if best is None or t < best[0]:
                        best = (t, cfg)
                hit = best[1]
                type(self)._TUNE_CACHE[key] = hit
                type(self)._tune_measured.add(key)
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
            hit = type(self)._TUNE_CACHE.get(key)
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
                        t = self._timed_cold(runw, warm, type(self).tune_iters())
                        if best is None or t < best[0]:
                            best = (t, tk + (tuple(encs),))
                hit = best[1]
                type(self)._TUNE_CACHE[key] = hit
                type(self)._tune_measured.add(key)
            if hit is not None:
                forced["wgrad"] = [(hit[0], hit[1], enc) for enc in hit[4]]
        return forced
