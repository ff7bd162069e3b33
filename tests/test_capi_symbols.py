"""The C-ABI library builds for gfx950, loads, and exports every symbol include/thk.h
declares (no compute calls: there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "thk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(thk_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(thk):
    lib = ctypes.CDLL(thk._capi.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 45
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/thk.h but not exported by libthk.so"


def test_python_binding_covers_header(thk):
    assert sorted(thk._capi.SIGNATURES) == declared_symbols()
    lib = thk._capi.load()
    assert lib.thk_abi_version() == 1


def test_library_is_gfx950_only():
    """No dual paths: the fat binary carries exactly one device target, gfx950."""
    blob = open(os.path.join(ROOT, "token-hawk_amd", "libthk.so"), "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_null_context_is_rejected(thk):
    lib = thk._capi.load()
    assert lib.thk_sync(None) < 0
    assert lib.thk_last_error(None) == b"null context"
    assert lib.thk_buf_ptr(None) is None


def test_model_shape_bytes(thk):
    """Algorithmic bytes per token match SURVEY.md §8d / BASELINE.md."""
    s7, s13 = thk.LLAMA_7B, thk.LLAMA_13B
    assert s7.n_ff == 11008 and s13.n_ff == 13824
    assert s7.weight_bytes() == 13_214_154_752
    assert s7.bytes_per_token(512) == 13_753_139_200
    assert s13.weight_bytes() == 25_703_219_200
    assert s13.bytes_per_token(512) == 26_545_377_280
