"""C-ABI checks that need no GPU: the library loads, exports every symbol include/x3dhip.h
declares, and the ctypes table in x3dhip/_lib.py covers exactly those symbols."""
import os
import re
import subprocess

import pytest

from x3dhip import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "x3dhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(x3d_[a-z0-9_]+)\s*\(", src)))


def test_header_and_ctypes_table_agree():
    assert _header_functions() == sorted(_lib.SIGNATURES.keys())


def test_library_loads_and_exports_every_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    h = _lib.lib()
    assert h.x3d_abi_version() == _lib.ABI_VERSION
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (x3d_[a-z0-9_]+)", out))
    assert set(_header_functions()) <= exported
    # pure host-side helpers are callable without a GPU
    assert h.x3d_pw_tiles(1, 24, 54, 1000, 1) == 16 and h.x3d_pw_tiles(64, 24, 54, 100000, 1) == 391
    assert h.x3d_pw_tiles(8, 216, 96, 3136, 1) == 49
    assert h.x3d_ew_tiles(4097) == 3
    assert h.x3d_dw_tiles(8, 54, 56, 56) >= 1
    assert h.x3d_last_error() is not None


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.X3DHipError):
        _lib.lib()
