"""CPU suite: the C-ABI library builds for gfx950, loads, and exports every symbol the header declares
(no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from kokoro_ruslan_amd import build
    return build.build()


def _header_decls():
    src = open(os.path.join(ROOT, "include", "kokoro_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|const char \*)\s*(kk_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args == "void" else len([a for a in args.split(",") if a.strip()])
        decls[m.group(1)] = n
    return decls


def test_library_exports_every_declared_symbol(libpath):
    decls = _header_decls()
    assert len(decls) >= 39
    lib = ctypes.CDLL(libpath)
    for name in decls:
        assert hasattr(lib, name), f"{name} declared in include/kokoro_hip.h but not exported"
    lib.kk_abi_version.restype = ctypes.c_int
    assert lib.kk_abi_version() == 1
    lib.kk_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.kk_last_error(), bytes)


def test_python_binding_matches_header_arity(libpath):
    from kokoro_ruslan_amd import lib as kk
    decls = _header_decls()
    for name, args in kk.SIGNATURES.items():
        assert name in decls, f"binding for undeclared function {name}"
        assert len(args) == decls[name], f"{name}: binding has {len(args)} args, header {decls[name]}"
    missing = set(decls) - set(kk.SIGNATURES) - {"kk_abi_version", "kk_last_error"}
    assert not missing, f"header functions without a Python binding: {missing}"
    kk.load()


def test_cfg_struct_sizes(libpath):
    """ctypes mirrors of KkLossCfg / KkOptCfg must match the C layout."""
    import subprocess, tempfile, textwrap
    from kokoro_ruslan_amd import lib as kk
    code = textwrap.dedent('''
        #include <stdio.h>
        #include "kokoro_hip.h"
        int main(void) { printf("%zu %zu\\n", sizeof(KkLossCfg), sizeof(KkOptCfg)); return 0; }
    ''')
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c")
        open(src, "w").write(code)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        a, b = map(int, subprocess.check_output([exe]).split())
    assert ctypes.sizeof(kk.KkLossCfg) == a and ctypes.sizeof(kk.KkOptCfg) == b


def test_product_path_fails_loudly_without_library(monkeypatch, tmp_path):
    from kokoro_ruslan_amd import lib as kk
    monkeypatch.setattr(kk, "_lib", None)
    monkeypatch.setattr(kk, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        kk.load()
