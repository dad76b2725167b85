"""Rewrites of an INFERENCE plan's forward list: eval-mode BatchNorm folded into the producing convolution, the stem + max-pool as one
launch, the stage-1 bottlenecks as one launch each, the 1x1 head inside the last transposed convolution's epilogue.  A mixin of
``engine.Plan`` (split out of engine.py in round 6).  Reference: PoseResNet.forward, src/modeling/simplebaseline/pose_resnet.py:234-248."""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import FuseBwdDesc, FuseDesc, IgemmDesc, check
from .graph import Act, _Call, _Marker, _desc, _ptr, _taps_array


class InferRewrites:
    fuse_head = True             # test hook: False keeps the inference head as separate launches (A/B against _fuse_head)

    fuse_stem = True             # test hook: False keeps the inference stem as convolution + max-pool launches (A/B against _fuse_stem_pool)

    fuse_bottleneck = os.environ.get("LH_FUSE_BOTTLENECK", "1") != "0"    # False keeps the stage-1 bottlenecks of inference plans as three launches (A/B against _fuse_bottleneck)

    def _fuse_head(self, nd, pack, bias):
        """Inference plans: `final_layer(relu(bn(deconv(x))))` (pose_resnet.py:245-246) as ONE launch.  When this 1x1
        convolution produces the network output and its only input is the output of a transposed convolution whose
        BatchNorm + ReLU were folded into its epilogue, the head is applied to every tile of that launch while it is in
        LDS (lh_igemm_phases_head): the C-channel activation is never written.  Returns False when the pattern does not
        apply (training plans, HRNet's head, more than 256 channels, a tile other than 256 x 256 on offer)."""
        x, y, k, s, p = nd["x"], nd["y"], nd["k"], nd["s"], nd["p"]
        if not type(self).fuse_head or self.with_bwd or self.es != 2 or k != 1 or s != 1 or p != 0 or y.c_valid > 32 or x.c > 256:
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
        if not type(self).fuse_bottleneck or self.with_bwd or self.training or self.es != 2:
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
        if not type(self).fuse_stem or self.with_bwd or self.training or self.es != 2 or x.c != 64 or x.c_valid != 64:
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
