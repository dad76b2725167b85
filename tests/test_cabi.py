"""CPU: the C-ABI library loads and exports every symbol include/lighthand_hip.h declares;
argument validation works without touching a GPU."""
import ctypes as C
import os
import re

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
