#!/usr/bin/env python3
"""Drives the HOST side of the sanitizer build (make -C lighthand_amd/csrc asan: AddressSanitizer + UBSan on the
library's host code) on a CPU box: no kernel is launched, no GPU is needed.

  * argument validation of the compute entry points (null pointers, bad dtypes, descriptors that break a hard rule),
  * the planner entry points over the convolution table of SimpleBaseline-R18/R50 and HRNet-W32/W48 at the benchmark
    shapes (src/modeling/simplebaseline/pose_resnet.py:144-232, src/modeling/hrnet/pose_hrnet.py:139-265), forward and
    data-gradient forms, every dtype: lh_igemm_tile / _config / _candidates / _stats_rows, lh_wgrad_tile / _candidates /
    _slab_bytes / _workspace_bytes, lh_igemm_phases_rows, lh_bn_stats_rows / _slab_bytes, lh_fuse_bwd_workspace_bytes,
    lh_pack_weight size queries, lh_stem_conv_rows, lh_maxpool3x3s2_bwd_gated_rows, the small workspace queries.

Run by tests/test_host_logic.py::test_host_side_sanitizer_build_is_clean as
    LD_PRELOAD=<libclang_rt.asan> ASAN_OPTIONS=detect_leaks=0 LH_LIB_PATH=lighthand_amd/liblighthand_hip_asan.so python tools/asan_host_check.py
and prints `asan host check: N calls, clean` when nothing was reported (a report aborts the process)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lighthand_amd import _lib                                   # noqa: E402
from lighthand_amd.engine import _desc                           # noqa: E402

lib = _lib.load()
calls = 0


def conv_fwd(n, h, w, cin, cout, k, s):
    p = (k - 1) // 2
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    taps = [(r - p, q - p) for r in range(k) for q in range(k)]
    return _desc(n, h, w, cin, cin, ho, wo, s, s, cout, ho, wo, 1, 1, 0, 0, cout, taps), (ho, wo)


def conv_dgrad_s1(n, h, w, cin, cout, k):
    p = (k - 1) // 2
    taps = [(p - r, p - q) for r in range(k) for q in range(k)]
    return _desc(n, h, w, cout, cout, h, w, 1, 1, cin, h, w, 1, 1, 0, 0, cin, taps)


def deconv_phases(n, h, w, cin, cout):
    """4x4 / stride 2 / pad 1 transposed convolution as its four sub-pixel phases (2x2 taps each)."""
    out = []
    for ph in range(2):
        for pw in range(2):
            taps = [(dh, dw) for dh in ((0, -1) if ph == 0 else (1, 0)) for dw in ((0, -1) if pw == 0 else (1, 0))]
            out.append(_desc(n, h, w, cin, cin, h, w, 1, 1, cout, 2 * h, 2 * w, 2, 2, ph, pw, cout, taps))
    return out


def layer_table():
    t = []
    for n, res in ((64, 256), (8, 256), (256, 384)):                      # C2 / C1 / C5 batch and size
        h = res // 4
        t += [(n, h, h, 64, 64, 1, 1), (n, h, h, 64, 64, 3, 1), (n, h, h, 64, 256, 1, 1), (n, h, h, 256, 64, 1, 1),
              (n, h, h, 256, 128, 1, 1), (n, h, h, 128, 128, 3, 2), (n, h, h, 256, 512, 1, 2),
              (n, h // 2, h // 2, 128, 512, 1, 1), (n, h // 2, h // 2, 512, 128, 1, 1), (n, h // 2, h // 2, 128, 128, 3, 1),
              (n, h // 2, h // 2, 512, 256, 1, 1), (n, h // 2, h // 2, 256, 256, 3, 2), (n, h // 4, h // 4, 256, 1024, 1, 1),
              (n, h // 4, h // 4, 1024, 256, 1, 1), (n, h // 4, h // 4, 256, 256, 3, 1), (n, h // 4, h // 4, 1024, 512, 1, 1),
              (n, h // 4, h // 4, 512, 512, 3, 2), (n, h // 8, h // 8, 512, 2048, 1, 1), (n, h // 8, h // 8, 2048, 512, 1, 1),
              (n, h // 8, h // 8, 512, 512, 3, 1), (n, h, h, 256, 21, 1, 1),
              (n, h, h, 64, 64, 3, 1), (n, h // 2, h // 2, 128, 128, 3, 1)]                    # R18 BasicBlocks
    for width in (32, 48):                                                                      # HRNet branches at batch 32
        for i in range(4):
            c, h = width << i, 64 >> i
            t += [(32, h, h, c, c, 3, 1)]
            if i:
                t += [(32, 64 >> (i - 1), 64 >> (i - 1), width << (i - 1), c, 3, 2), (32, h, h, c, width, 1, 1)]
    t += [(3, 7, 5, 40, 24, 3, 1), (2, 9, 11, 8, 136, 1, 1), (1, 5, 5, 264, 72, 3, 2)]         # ragged shapes
    return t


def run():
    global calls
    ibuf = (C.c_int * (5 * 320))()
    wbuf = (C.c_int * (8 * 320))()
    a, b, c, d4 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    for (n, h, w, cin, cout, k, s) in layer_table():
        fwd, (ho, wo) = conv_fwd(n, h, w, cin, cout, k, s)
        descs = [fwd] + ([conv_dgrad_s1(n, h, w, cin, cout, k)] if s == 1 else [])
        for d in descs:
            for dt in (_lib.LH_F32, _lib.LH_BF16, _lib.LH_F16):
                lib.lh_igemm_tile(C.byref(d), dt, C.byref(a), C.byref(b), C.byref(c))
                cfg = (C.c_int * 5)()
                lib.lh_igemm_config(C.byref(d), dt, cfg)
                nc = lib.lh_igemm_candidates(C.byref(d), dt, ibuf, 320)
                assert 0 <= nc < 320, nc
                lib.lh_igemm_candidates(C.byref(d), dt, ibuf, 1)              # a buffer that is too small must be respected
                lib.lh_igemm_stats_rows(C.byref(d), dt)
                for i in range(min(nc, 320)):                                 # every candidate resolves to a valid configuration
                    d.cfg[0], d.cfg[1], d.cfg[2], d.cfg[3] = ibuf[5 * i], ibuf[5 * i + 1], ibuf[5 * i + 2], ibuf[5 * i + 3]
                    lib.lh_igemm_config(C.byref(d), dt, cfg)
                    rows = lib.lh_igemm_stats_rows(C.byref(d), dt)
                    for nterms in (1, 2):                                     # the gated launch's slab (round 6): as many rows, except on the pointwise kernel
                        gr = lib.lh_igemm_gated_rows(C.byref(d), dt, nterms)
                        assert gr >= 1 and (cfg[2] == 1 or gr == rows), (gr, rows, tuple(cfg))
                    calls += 4
                for j in range(8):
                    d.cfg[j] = 0
                calls += 5
        for dt in (_lib.LH_F32, _lib.LH_BF16, _lib.LH_F16):
            lib.lh_wgrad_tile(C.byref(fwd), cout, cin, dt, C.byref(a), C.byref(b), C.byref(c), C.byref(d4))
            nw = lib.lh_wgrad_candidates(C.byref(fwd), cout, cin, dt, wbuf, 320)
            assert 0 <= nw <= 320, nw
            lib.lh_wgrad_slab_bytes(C.byref(fwd), cout, cin, dt)
            lib.lh_wgrad_workspace_bytes(C.byref(fwd), cout, cin, dt)
            calls += 4
        lib.lh_bn_stats_rows(n * ho * wo, cout)
        lib.lh_bn_stats_slab_bytes(lib.lh_bn_stats_rows(n * ho * wo, cout), cout)
        lib.lh_fuse_bwd_workspace_bytes(n, ho, wo, cout)
        calls += 4
        # pack size query (out = NULL): OIHW weights of this layer
        sz = C.c_size_t()
        taps = (C.c_int * (2 * k * k))(*[v for r in range(k) for q in range(k) for v in (r, q)])
        lib.lh_pack_weight(None, None, C.byref(sz), cout, cin, cin * k * k, k * k, k, 1, k * k, taps, _lib.LH_BF16, None)
        calls += 1
    for (n, h, w, cin, cout) in ((64, 8, 8, 2048, 256), (64, 16, 16, 256, 256), (256, 48, 48, 256, 256), (2, 5, 7, 24, 40)):
        ds = deconv_phases(n, h, w, cin, cout)
        arr = (C.POINTER(_lib.IgemmDesc) * 4)(*[C.pointer(x) for x in ds])
        for dt in (_lib.LH_F32, _lib.LH_BF16, _lib.LH_F16):
            lib.lh_igemm_phases_rows(arr, 4, dt)
            calls += 1
    # table launches of the weight gradient (lh_wgrad_table_build, size query: host arithmetic only -- the per-problem split counts, the
    # XCD-aware work-item order with its coverage check, blob offsets): the layers of every batch / size of the table as ONE table each,
    # every compiled-in tile class, automatic and forced item lengths, both item orders
    by_shape = {}
    for (n, h, w, cin, cout, k, st) in layer_table():
        if cin % 8 == 0 and cout % 8 == 0:
            by_shape.setdefault((n, h * w > 0), []).append((n, h, w, cin, cout, k, st))
    for layers in by_shape.values():
        keep, arr = [], (_lib.WgradCall * len(layers))()
        for i, (n, h, w, cin, cout, k, st) in enumerate(layers):
            fwd, _ = conv_fwd(n, h, w, cin, cout, k, st)
            taps = (C.c_int * (2 * k * k))(*[v for r in range(k) for q in range(k) for v in (r, q)])
            keep += [fwd, taps]
            arr[i].d, arr[i].rows, arr[i].x, arr[i].dy, arr[i].dy_pix_stride = C.pointer(fwd), 0, 4096, 8192, cout
            arr[i].n_out, arr[i].n_in, arr[i].grad = cout, cin, 1 << 20
            arr[i].so, arr[i].si, arr[i].sr, arr[i].ss = cin * k * k, k * k, k, 1
            arr[i].taps_rs, arr[i].accumulate = C.cast(taps, C.POINTER(C.c_int)), 0
        info = _lib.WgradTableInfo()
        for xcd in ("1", "0"):
            os.environ["LH_WGRAD_TABLE_XCD"] = xcd
            for cfg in ((256, 256, 64, 2), (128, 128, 64, 3), (64, 64, 32, 4)):
                for target in (0, 1, 7, 64, 100000):
                    rc = lib.lh_wgrad_table_build(arr, len(layers), _lib.LH_BF16, (C.c_int * 4)(*cfg), target, None, None, 0, C.byref(info))
                    assert rc == 0 and info.n_items > 0 and info.table_bytes >= info.off_fold_items >= info.off_fold_args > info.off_items > 0, (rc, lib.lh_last_error())
                    calls += 1
        os.environ.pop("LH_WGRAD_TABLE_XCD", None)
    assert lib.lh_wgrad_table_build(None, 1, _lib.LH_BF16, (C.c_int * 4)(128, 128, 64, 3), 0, None, None, 0, C.byref(_lib.WgradTableInfo())) != 0
    for args in ((64, 128, 128), (8, 128, 128), (3, 37, 41)):
        lib.lh_stem_conv_rows(*args)
        lib.lh_maxpool3x3s2_bwd_gated_rows(args[0], args[1], args[2], 64, _lib.LH_BF16)
        calls += 2
    lib.lh_mse_workspace_bytes(64 * 21 * 64 * 64)
    lib.lh_channel_sum_workspace_bytes(256)
    lib.lh_image_jitter_workspace_bytes(64)
    lib.lh_pack_chunk_elems()
    [lib.lh_dtype_size(i) for i in range(-1, 5)]
    calls += 10
    # argument validation: every one of these must return a status and set the error text, never touch memory
    d = _lib.IgemmDesc()
    assert lib.lh_igemm(C.byref(d), None, None, None, None, None, None, None, None, None, _lib.LH_BF16, None) != 0
    assert lib.lh_igemm(None, None, None, None, None, None, None, None, None, None, 7, None) != 0
    fwd, _ = conv_fwd(4, 16, 16, 64, 64, 3, 1)
    fwd.cfg[0], fwd.cfg[1], fwd.cfg[2], fwd.cfg[3] = 96, 100, 3, 128                    # no such tile
    assert lib.lh_igemm_config(C.byref(fwd), _lib.LH_BF16, (C.c_int * 5)()) != 0
    assert lib.lh_fuse_fwd(None, None, 1, 1, 1, 8, _lib.LH_BF16, None) != 0
    assert lib.lh_fuse_bwd(None, 1, 1, 1, 8, None, _lib.LH_BF16, None) != 0
    assert lib.lh_adam_step(None, None, None, None, 16, None, None, None, 1e-3, None) != 0
    assert lib.lh_heatmap_argmax(None, 1, 4, 4, 4.0, None, None, None, None) != 0
    assert lib.lh_igemm_multi(None, 2, _lib.LH_BF16, None) != 0
    assert lib.lh_wgrad_fused_multi(None, 2, _lib.LH_BF16, None) != 0
    bd = _lib.BottleneckDesc(2, 16, 16, 256, 64, 256)
    assert lib.lh_bottleneck_infer(C.byref(bd), None, None, None, None, None, None, None, None, None, None, None, None, _lib.LH_F16, None) != 0
    bd.mid = 128                                                                        # not the stage this kernel implements
    one = C.c_void_p(16)
    assert lib.lh_bottleneck_infer(C.byref(bd), one, one, one, one, one, one, one, one, one, one, one, C.c_void_p(32), _lib.LH_F16, None) != 0
    assert len(lib.lh_last_error()) > 0
    calls += 12


if __name__ == "__main__":
    run()
    print(f"asan host check: {calls} calls, clean")
