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
    from kokoro_ruslan_amd import lib as kk
    assert lib.kk_abi_version() == 2 == kk.ABI_VERSION
    lib.kk_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.kk_last_error(), bytes)


def test_python_binding_matches_header_arity(libpath):
    from kokoro_ruslan_amd import lib as kk
    decls = _header_decls()
    for name, args in kk.SIGNATURES.items():
        assert name in decls, f"binding for undeclared function {name}"
        assert len(args) == decls[name], f"{name}: binding has {len(args)} args, header {decls[name]}"
    # (entry points without a stream argument are bound by hand in lib.load())
    missing = set(decls) - set(kk.SIGNATURES) - {"kk_abi_version", "kk_last_error", "kk_last_kernel", "kk_seg_sumsq_rec_capacity", "kk_attn_warm_next"}
    lib = kk.load()
    assert lib.kk_seg_sumsq_rec_capacity() >= 2048 and lib.kk_seg_sumsq_rec_offset() > 0
    assert not missing, f"header functions without a Python binding: {missing}"
    kk.load()


def test_cfg_struct_sizes(libpath):
    """ctypes mirrors of the C structs (cfg structs, descriptor tables, the encoder-stack descriptor) must match the C layout:
    sizes and the offsets of a few fields deep inside."""
    import subprocess, tempfile, textwrap
    from kokoro_ruslan_amd import lib as kk
    code = textwrap.dedent('''
        #include <stdio.h>
        #include <stddef.h>
        #include "kokoro_hip.h"
        int main(void) {
            printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(KkLossCfg), sizeof(KkOptCfg), sizeof(KkWgradDesc),
                   sizeof(KkReduceDesc), sizeof(KkAttnHeadNorm), sizeof(KkEncLayer), sizeof(KkEncStack), offsetof(KkEncLayer, next_y_bf16),
                   offsetof(KkEncLayer, dpr), offsetof(KkEncStack, placement), offsetof(KkEncStack, layer));
            return 0;
        }
    ''')
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c")
        open(src, "w").write(code)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        got = list(map(int, subprocess.check_output([exe]).split()))
    want = [ctypes.sizeof(kk.KkLossCfg), ctypes.sizeof(kk.KkOptCfg), ctypes.sizeof(kk.KkWgradDesc), ctypes.sizeof(kk.KkReduceDesc),
            ctypes.sizeof(kk.KkAttnHeadNorm), ctypes.sizeof(kk.KkEncLayer), ctypes.sizeof(kk.KkEncStack), kk.KkEncLayer.next_y_bf16.offset,
            kk.KkEncLayer.dpr.offset, kk.KkEncStack.placement.offset, kk.KkEncStack.layer.offset]
    assert got == want, (got, want)
    assert kk.KK_ENC_MAX_LAYERS == 8 and ctypes.sizeof(kk.KkEncStack) < 4096, "the descriptor travels as a kernel argument"


def test_encoder_stack_descriptor_builder(libpath):
    """lib.enc_stack fills every pointer field from tensors and leaves the optional ones null (CPU tensors: no launch here)."""
    import torch
    from kokoro_ruslan_amd import lib as kk
    t = lambda *s: torch.zeros(*s)
    layer = {n: t(4) for n, _ in kk.KkEncLayer._fields_[:34]}
    layer.update(next_y_bf16=1, site=1000, p=0.15, dpr=0.1)
    d = kk.enc_stack(2, 33, 512, 1536, 8, None, t(64), t(64), None, torch.zeros(512, dtype=torch.int32), [layer] * 6, placement=1)
    assert (d.B, d.S, d.H, d.F, d.heads, d.layers, d.placement) == (2, 33, 512, 1536, 8, 6, 1)
    assert d.key_mask is None and d.seed is None and d.trace is None
    assert d.layer[5].w_qkv == layer["w_qkv"].data_ptr() and d.layer[5].site == 1000 and abs(d.layer[5].dpr - 0.1) < 1e-7
    assert d.layer[6].w_qkv is None
    lib = kk.load()
    wgs = lib.kk_encoder_stack_workgroups()          # follows the device's CU count: 0 without a GPU, 256 on an MI355X
    assert wgs in (0, 256) or (wgs % 8 == 0 and 0 < wgs < 256)
    assert lib.kk_encoder_stack_supported(8, 64, 512, 1536, 8, 6) == (1 if wgs == 256 else 0) or wgs not in (0, 256)
    assert lib.kk_encoder_stack_supported(8, 129, 512, 1536, 8, 6) == 0 and lib.kk_encoder_stack_supported(8, 64, 512, 2048, 8, 6) == 0


def test_product_path_fails_loudly_without_library(monkeypatch, tmp_path):
    from kokoro_ruslan_amd import lib as kk
    monkeypatch.setattr(kk, "_lib", None)
    monkeypatch.setattr(kk, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        kk.load()
