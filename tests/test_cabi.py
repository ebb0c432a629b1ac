"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
declared in include/a3t_hip.h (no compute calls: there is no GPU in the build container)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "a3t_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(a3t_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from a3t_amd import _lib, build
    build.build(verbose=False)
    lib = _lib.load()
    decl = _declared()
    assert len(decl) >= 30
    for name in decl:
        assert hasattr(lib, name), name
    assert sorted(_lib.EXPORTS) == decl
    assert b"gfx950" in lib.a3t_version()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from a3t_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.A3TLibraryError):
        _lib.load()


def test_gemm_desc_layout_matches_header():
    import ctypes
    from a3t_amd._lib import GemmDesc
    # 6 pointers, 3 i32 (+pad), 6 i64, 2 i32, 6 i64, 5 i32 + f32 + 3 i32 + 4 i32 -> natural C layout
    assert GemmDesc.A.offset == 0 and GemmDesc.M.offset == 48
    assert GemmDesc.a_rs.offset == 64 and GemmDesc.c_rs.offset == 104
    assert GemmDesc.batch.offset == 112 and GemmDesc.a_bs0.offset == 120
    assert GemmDesc.taps.offset == 168 and GemmDesc.alpha.offset == 188
    assert GemmDesc.s_dtype.offset == 220 and GemmDesc.colsum.offset == 232 and GemmDesc.drop_p.offset == 256
    assert GemmDesc.keep_out.offset == 264 and GemmDesc.keep_in.offset == 272 and ctypes.sizeof(GemmDesc) == 280
