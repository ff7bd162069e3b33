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


def test_overlap_code_object_holds_the_default_geometry():
    """libthk_ovl.hsaco (kernels of the overlapped dispatch, loaded by thk_ovl.cpp with the HSA runtime) is built next to libthk.so
    and holds, in all four flavours, the kernels the default launch geometry of LLaMA-7B and 13B asks for (the names are composed
    by the launchers in thk_kernels.hip; a missing one would only show as an error on the GPU)."""
    import subprocess
    import pytest
    import __graft_entry__ as graft
    graft.build_libthk()
    hsaco = os.path.join(graft.PKG_DIR, "libthk_ovl.hsaco")
    assert os.path.exists(hsaco)
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not in this image")
    syms = subprocess.run([readelf, "-s", "--wide", hsaco], capture_output=True, text=True, check=True).stdout
    want = ["thk_ovl_gemv_2_8_8_3_2_0_1", "thk_ovl_gemv_2_8_8_1_2_0_1", "thk_ovl_gemv_1_8_8_2_1_4_1", "thk_ovl_gemv_2_8_8_1_3_0_1",
            "thk_ovl_gemv_1_22_22_0_1_0_0", "thk_ovl_gemv_1_8_8_1_4_0_1",                                  # 7B: qkv (+embedding fold), wo, w1|w3, w2, lm-head
            "thk_ovl_gemv_2_10_10_3_2_0_0", "thk_ovl_gemv_2_10_10_1_2_0_0", "thk_ovl_gemv_1_10_10_2_1_4_1", "thk_ovl_gemv_2_10_10_1_3_0_1",
            "thk_ovl_gemv_1_27_27_0_1_0_1", "thk_ovl_gemv_2_5_10_1_4_0_0",                                 # 13B
            "thk_ovl_attn_128_8_0"]
    for base in want:
        for f in range(4):
            assert f"{base}_f{f}.kd" in syms, f"{base}_f{f}"
    for name in ("thk_ovl_finish_token_f0.kd", "thk_ovl_finish_token_f1.kd", "thk_ovl_batch_begin.kd", "thk_ovl_batch_end.kd"):
        assert name in syms, name
