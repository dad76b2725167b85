"""CPU: the C-ABI library loads and exports every symbol include/lighthand_hip.h declares;
argument validation works without touching a GPU."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "lighthand_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from lighthand_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but missing from the library"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes prototype"
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.lh_version() >= 100
    assert [lib.lh_dtype_size(i) for i in range(4)] == [4, 2, 2, 0]


def test_ctypes_prototypes_have_the_header_arity():
    """Every prototype of include/lighthand_hip.h has as many parameters as its ctypes mirror in _lib.SIGNATURES, and pointer /
    integer / floating parameters sit at the same positions."""
    from lighthand_amd import _lib
    text = open(os.path.join(ROOT, "include", "lighthand_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = dict(re.findall(r"\b(lh_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S))
    assert set(protos) == set(_lib.SIGNATURES)
    for name, args in protos.items():
        params = [a.strip() for a in args.split(",")] if args.strip() not in ("", "void") else []
        want = _lib.SIGNATURES[name][1]
        assert len(params) == len(want), (name, params, want)
        for prm, ct in zip(params, want):
            is_ptr = "*" in prm
            c_ptr = ct in (C.c_void_p, C.c_char_p) or isinstance(ct, type) and issubclass(ct, C._Pointer)
            assert is_ptr == c_ptr, (name, prm, ct)
            if not is_ptr:
                floating = bool(re.match(r"(const\s+)?(float|double)\b", prm))
                assert floating == (ct in (C.c_float, C.c_double)), (name, prm, ct)


def test_argument_validation_sets_error_text():
    from lighthand_amd import _lib
    lib = _lib.load()
    d = _lib.IgemmDesc()
    rc = lib.lh_igemm(C.byref(d), None, None, None, None, None, None, None, None, None, _lib.LH_BF16, None)
    assert rc == -1 and b"null" in lib.lh_last_error()
    try:
        _lib.check(rc, "lh_igemm")
    except _lib.LightHandError as e:
        assert "lh_igemm" in str(e)
    else:
        raise AssertionError("check() must raise")
    nbytes = C.c_size_t(0)
    taps = (C.c_int * 2)(0, 0)
    assert lib.lh_pack_weight(None, None, C.byref(nbytes), 21, 256, 256, 1, 0, 0, 1, taps, _lib.LH_BF16, None) == 0
    assert nbytes.value == 128 * 256 * 2            # rows padded to 128, K padded to the 128-byte step


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from lighthand_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.load()
    except _lib.LightHandError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load() must fail when the extension is missing")


def _conv_desc(cin, cout, hw=16, k=1):
    from lighthand_amd import _lib
    d = _lib.IgemmDesc()
    d.n, d.hi, d.wi, d.in_pix_stride, d.k_run = 1, hw, hw, cin, cin
    d.ho, d.wo, d.sh, d.sw, d.cout = hw, hw, 1, 1, cout
    d.OH, d.OW, d.osh, d.osw, d.ooh, d.oow, d.out_pix_stride = hw, hw, 1, 1, 0, 0, cout
    d.ntaps = k * k
    for t in range(k * k):
        d.dh[t], d.dw[t] = t // k - k // 2, t % k - k // 2
    return d


def test_explicit_conv_configuration_must_fit():
    """An explicit lh_igemm_desc.cfg that does not pass the rules of lh_igemm_candidates is refused with LH_ERR_ARG
    instead of launching (a 256-row tile on a weight pack with an odd number of 128-row blocks would read past it)."""
    from lighthand_amd import _lib
    lib = _lib.load()
    d = _conv_desc(256, 128)
    cfg = (C.c_int * 5)()
    d.cfg[0], d.cfg[1], d.cfg[2], d.cfg[3] = 256, 256, 3, 64          # compiled in, but cout = 128 is ONE 128-row block
    assert lib.lh_igemm_config(C.byref(d), _lib.LH_BF16, cfg) == -1 and b"does not fit" in lib.lh_last_error()
    d.cfg[0], d.cfg[1], d.cfg[2], d.cfg[3] = 128, 128, 2, 128
    assert lib.lh_igemm_config(C.byref(d), _lib.LH_BF16, cfg) == 0 and tuple(cfg[:4]) == (128, 128, 2, 128)
    d.cfg[0], d.cfg[1], d.cfg[2], d.cfg[3] = 256, 16, 1, 512          # pointwise panel with the wrong K padding
    assert lib.lh_igemm_config(C.byref(d), _lib.LH_BF16, cfg) == -1
    d.cfg[0], d.cfg[1], d.cfg[2], d.cfg[3] = 128, 16, 1, 256
    assert lib.lh_igemm_config(C.byref(d), _lib.LH_BF16, cfg) == 0 and cfg[2] == 1
    assert lib.lh_igemm_config(C.byref(d), _lib.LH_F32, cfg) == -1    # 16-bit types only
    d3 = _conv_desc(256, 128, k=3)
    d3.cfg[0], d3.cfg[1], d3.cfg[2], d3.cfg[3] = 128, 16, 1, 256      # not a 1x1 form
    assert lib.lh_igemm_config(C.byref(d3), _lib.LH_BF16, cfg) == -1
    # every candidate the library offers resolves, and the pointwise ones report their slab rows
    buf = (C.c_int * (5 * 64))()
    d = _conv_desc(64, 256, hw=64)
    n = lib.lh_igemm_candidates(C.byref(d), _lib.LH_BF16, buf, 64)
    kinds = set()
    for i in range(n):
        for j in range(4):
            d.cfg[j] = buf[5 * i + j]
        assert lib.lh_igemm_config(C.byref(d), _lib.LH_BF16, cfg) == 0
        rows = lib.lh_igemm_stats_rows(C.byref(d), _lib.LH_BF16)
        kinds.add(d.cfg[2] == 1)
        assert rows == (64 * 64 + d.cfg[1] - 1) // d.cfg[1] if d.cfg[2] != 1 else rows in (8 * k for k in range(1, 65))
    assert kinds == {True, False}


def test_explicit_wgrad_configuration_must_fit():
    from lighthand_amd import _lib
    lib = _lib.load()
    d = _conv_desc(64, 64, k=3)
    bo, bi, ns, ring = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    buf = (C.c_int * (5 * 128))()
    n = lib.lh_wgrad_candidates(C.byref(d), 64, 64, _lib.LH_BF16, buf, 128)
    assert n >= 1
    for i in range(n):                                                 # every offered plan resolves
        d.cfg[5], d.cfg[6], d.cfg[7] = buf[5 * i], buf[5 * i + 1], buf[5 * i + 2]
        assert lib.lh_wgrad_tile(C.byref(d), 64, 64, _lib.LH_BF16, C.byref(bo), C.byref(bi), C.byref(ns), C.byref(ring)) == 0
    big = _conv_desc(256, 256, k=3)
    n = lib.lh_wgrad_candidates(C.byref(big), 256, 256, _lib.LH_BF16, buf, 128)
    enc = next(buf[5 * i + 2] for i in range(n) if (buf[5 * i], buf[5 * i + 1]) == (256, 256))
    d.cfg[5], d.cfg[6], d.cfg[7] = 256, 256, enc                       # a compiled-in 256 x 256 plan on a 64 x 64 gradient
    assert lib.lh_wgrad_tile(C.byref(d), 64, 64, _lib.LH_BF16, C.byref(bo), C.byref(bi), C.byref(ns), C.byref(ring)) == -1
    assert b"does not fit" in lib.lh_last_error()


def test_wgrad_table_planner_runs_without_a_device(monkeypatch):
    """lh_wgrad_table_build's size query is host arithmetic: the weight gradients of R50's stage 4 at batch 64 (three bottlenecks +
    the projection, pose_resnet.py:61-99, 177-192: M = 64 x 8 x 8 pixels) in one table.  Automatic item length: the tiles alone fill the
    machine twice -> split-free, the 1x1 members written by the kernel itself (only the 3x3 ones need the fold launch); a short item
    length splits every member; an unknown tile is refused."""
    from lighthand_amd import _lib
    lib = _lib.load()
    monkeypatch.setenv("LH_WGRAD_TABLE_XCD", "0")                  # the plain longest-first item order: exact item counts
    shapes = [(512, 2048, 1), (512, 512, 3), (2048, 512, 1)] * 2 + [(512, 1024, 1), (512, 512, 3), (2048, 512, 1), (2048, 1024, 1)]
    descs, rs, calls = [], [], (_lib.WgradCall * len(shapes))()
    for i, (cout, cin, k) in enumerate(shapes):
        d = _conv_desc(cin, cout, hw=8, k=k)
        d.n = 64
        taps = (C.c_int * (2 * k * k))(*[v for r in range(k) for q in range(k) for v in (r, q)])
        descs.append(d); rs.append(taps)
        calls[i].d, calls[i].rows, calls[i].x, calls[i].dy, calls[i].dy_pix_stride = C.pointer(d), 0, 4096, 8192, cout
        calls[i].n_out, calls[i].n_in, calls[i].grad = cout, cin, 1 << 20
        calls[i].so, calls[i].si, calls[i].sr, calls[i].ss = cin * k * k, k * k, k, 1
        calls[i].taps_rs, calls[i].accumulate = C.cast(taps, C.POINTER(C.c_int)), 0
    info = _lib.WgradTableInfo()
    cfg = (C.c_int * 4)(128, 128, 64, 3)
    assert lib.lh_wgrad_table_build(calls, len(shapes), _lib.LH_BF16, cfg, 0, None, None, 0, C.byref(info)) == 0, lib.lh_last_error()
    tiles = sum((co // 128) * (ci // 128) * k * k for co, ci, k in shapes)
    assert (info.n_problems, info.n_items, info.nsplit_max, info.n_fold) == (len(shapes), tiles, 1, 3)
    assert info.target_stages == 64 * 8 * 8 // 64 and info.n_fold_items > 0
    assert info.workspace_bytes == 3 * 512 * 512 * 9 * 4 and info.table_bytes > info.off_fold_items > info.off_fold_args > info.off_items > 0
    assert lib.lh_wgrad_table_build(calls, len(shapes), _lib.LH_BF16, cfg, 8, None, None, 0, C.byref(info)) == 0
    assert info.nsplit_max == 8 and info.n_items == 8 * tiles and info.n_fold == len(shapes)
    big = (C.c_int * 4)(256, 256, 32, 3)
    assert lib.lh_wgrad_table_build(calls, len(shapes), _lib.LH_BF16, big, 0, None, None, 0, C.byref(info)) == 0
    assert 3 * 256 < info.n_items < 6 * 256 and info.nsplit_max >= 2   # 236 tiles do not fill 256 CUs twice: about four rounds of items
    # the default item order lays the taps x tiles of one pixel split of one layer on ONE XCD, eight groups per round, rounds padded
    # with empty items: never fewer items than the plain order, and the padding stays below one round's worth per round
    monkeypatch.delenv("LH_WGRAD_TABLE_XCD")
    plain = info.n_items
    assert lib.lh_wgrad_table_build(calls, len(shapes), _lib.LH_BF16, big, 0, None, None, 0, C.byref(info)) == 0
    assert plain <= info.n_items <= 2 * plain
    bad = (C.c_int * 4)(96, 96, 64, 3)
    assert lib.lh_wgrad_table_build(calls, len(shapes), _lib.LH_BF16, bad, 0, None, None, 0, C.byref(info)) == -3
    assert b"not compiled in" in lib.lh_last_error()
    assert lib.lh_wgrad_table_build(calls, len(shapes), _lib.LH_F32, cfg, 0, None, None, 0, C.byref(info)) == -1


def test_ctypes_structs_mirror_the_header_layout(tmp_path):
    """The ctypes mirrors in lighthand_amd/_lib.py must have the size and the field offsets of the structs that
    include/lighthand_hip.h declares: a probe compiled with gcc against the header prints sizeof / offsetof of every field
    (a field added on one side only would shift every later argument silently)."""
    import ctypes as C
    import shutil
    import subprocess
    from lighthand_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    pairs = {"lh_igemm_desc": _lib.IgemmDesc, "lh_fuse_desc": _lib.FuseDesc, "lh_fuse_bwd_desc": _lib.FuseBwdDesc,
             "lh_igemm_call": _lib.IgemmCall, "lh_wgrad_call": _lib.WgradCall, "lh_fuse_fwd_call": _lib.FuseFwdCall,
             "lh_fuse_bwd_call": _lib.FuseBwdCall, "lh_bn_finalize_call": _lib.BnFinalizeCall, "lh_head": _lib.Head,
             "lh_pack_item": _lib.PackItem, "lh_pack_out": _lib.PackOut, "lh_pack_conv": _lib.PackConv, "lh_bn_bwd_gate": _lib.BnBwdGate,
             "lh_bottleneck_desc": _lib.BottleneckDesc, "lh_wgrad_table_info": _lib.WgradTableInfo}
    rename = {"in_": "in", "pad_": None}                       # ctypes-side spellings; None = padding without a C name
    lines = []
    for cname, cls in pairs.items():
        lines.append(f'printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, *_ in cls._fields_:
            cf = rename.get(fname, fname)
            if cf is not None:
                lines.append(f'printf("{cname} {fname} %zu\\n", offsetof({cname}, {cf}));')
    src = tmp_path / "probe.c"
    src.write_text("#include <stdio.h>\n#include <stddef.h>\n#include \"lighthand_hip.h\"\nint main(void) {\n" + "\n".join(lines) + "\nreturn 0; }\n")
    exe = tmp_path / "probe"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe)], check=True)
    got = {}
    for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        cname, field, val = line.split()
        got[(cname, field)] = int(val)
    for cname, cls in pairs.items():
        assert got[(cname, "size")] == C.sizeof(cls), (cname, got[(cname, "size")], C.sizeof(cls))
        for fname, *_ in cls._fields_:
            if rename.get(fname, fname) is not None:
                assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)

