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
    assert GemmDesc.keep_out.offset == 264 and GemmDesc.keep_in.offset == 272
    # (round 6: the fused-LayerNorm block of round 4 left the descriptor with its kernel, tools/experiments/ln_fwd_in_panel_epilogue.patch)
    # (round 6: + a_signmask, the sign-tagged probabilities of the attention backward's dV product)
    assert GemmDesc.a_signmask.offset == 280 and GemmDesc.keep_layout.offset == 284 and GemmDesc.A2.offset == 288
    # (... and the second product of one streaming launch: A2, B2, b2_cs, b2_bs0, b2_bs1, colsum2)
    assert GemmDesc.colsum2.offset == 328 and GemmDesc.a2_rs.offset == 336 and GemmDesc.a_unaligned.offset == 344
    assert ctypes.sizeof(GemmDesc) == 352


def test_gemm_desc_layout_as_the_c_compiler_sees_the_header(tmp_path):
    """offsetof / sizeof of a3t_gemm_desc from include/a3t_hip.h compiled by gcc against the ctypes mirror, field by field."""
    import ctypes
    import shutil
    import subprocess
    from a3t_amd._lib import GemmDesc
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    names = [f[0] for f in GemmDesc._fields_]
    src = tmp_path / "off.c"
    body = "\n".join(f'    printf("{n} %zu\\n", offsetof(a3t_gemm_desc, {n}));' for n in names)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "a3t_hip.h"\nint main(void) {\n' + body +
                   '\n    printf("sizeof %zu\\n", sizeof(a3t_gemm_desc));\n    return 0;\n}\n')
    exe = tmp_path / "off"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run(["gcc", "-I", inc, str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for n in names:
        assert int(out[n]) == getattr(GemmDesc, n).offset, n
    assert int(out["sizeof"]) == ctypes.sizeof(GemmDesc)
