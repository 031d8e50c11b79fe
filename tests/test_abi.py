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
    assert h.x3d_dw_tiles(8, 54, 16, 56, 56) == 4                     # 4 row tiles of 14 rows, one T segment
    assert h.x3d_dw_tiles(8, 432, 16, 7, 7) == 2                      # 216 workgroups < 256 CUs: two T segments
    assert h.x3d_last_error() is not None


def test_option_table_set_get_reset():
    """x3d_set_option / x3d_get_option / x3d_reset_options (host only): every declared name round-trips, values are range
    checked, unknown names fail, and tile-count queries follow an option at call time."""
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    names = _lib.option_names()
    assert {"fb_grid", "bwd_terms", "dgrad_f32", "wgrad_f32", "dw_th", "no_pw6", "no_pw7", "no_pwfs", "dw_tsplit_wgs", "dw_tsplit_wgs_fwd",
            "pw_two_tiles_k"} <= set(names)
    assert _lib.get_option("bwd_terms") == 3 and _lib.get_option("fb_grid") == 512
    h = _lib.lib()
    assert h.x3d_pw_bwd_fused_groups(2, 24964) == 512
    with _lib.options(fb_grid=7, bwd_terms=2):
        assert _lib.get_option("fb_grid") == 7 and _lib.get_option("bwd_terms") == 2
        assert h.x3d_pw_bwd_fused_groups(2, 24964) == 7
    assert _lib.get_option("fb_grid") == 512 and _lib.get_option("bwd_terms") == 3
    for bad in (("bwd_terms", 4), ("fb_grid", 0), ("dw_th", 17), ("dgrad_f32", 2)):
        with pytest.raises(_lib.X3DHipError):
            _lib.set_option(*bad)
    with pytest.raises(_lib.X3DHipError):
        _lib.set_option("no_such_option", 1)
    _lib.set_option("dw_th", 8)
    assert h.x3d_reset_options() == 0 and _lib.get_option("dw_th") == 16


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.X3DHipError):
        _lib.lib()
