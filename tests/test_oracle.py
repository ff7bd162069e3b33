"""CPU tests of the oracle (oracle/thk_oracle.c): golden fixtures, independent numpy
implementations, faithful-vs-fast agreement.  No GPU, no libthk compute."""
import os
import zlib

import numpy as np
import pytest

from ref_numpy import RefModel

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.npz"))


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


# ------------------------------------------------------------------ A17 fp16 <-> fp32
def test_fp16_to_fp32_all_patterns_match_ieee(orc):
    h = np.arange(65536, dtype=np.uint16)
    got = orc.fp16_to_fp32(h)
    ref = h.view(np.float16).astype(np.float32)
    fin = np.isfinite(ref)
    assert (got.view(np.uint32)[fin] == ref.view(np.uint32)[fin]).all()
    assert np.isnan(got[np.isnan(ref)]).all() and (got[np.isinf(ref)] == ref[np.isinf(ref)]).all()
    assert crc(got[fin].view(np.uint32)) == int(GOLD["fp16_all_finite_crc"])


def test_fp32_to_fp16_matches_golden(orc):
    got = orc.fp32_to_fp16(GOLD["fp32_samples"])
    assert (got == GOLD["fp32_to_fp16_expected"]).all()


def test_fp16_roundtrip_is_identity(orc):
    h = np.arange(65536, dtype=np.uint16)
    f = orc.fp16_to_fp32(h)
    fin = np.isfinite(f)
    assert (orc.fp32_to_fp16(f[fin]) == h[fin]).all()


# ------------------------------------------------------------------ synthetic generator
def test_synth_generator_matches_golden(orc):
    for key in GOLD.files:
        if key.startswith("synth16_head_"):
            name = key[len("synth16_head_"):]
            v = orc.synth_f16(name, orc.TENSOR_SEED, orc.TENSOR_SIGMA, 1 << 16)
            assert (v[:32] == GOLD[key]).all()
            assert crc(v) == int(GOLD["synth16_crc_" + name])
    g = orc.synth_gain("norm.weight", orc.TENSOR_SEED, orc.TENSOR_SIGMA, 4096)
    assert (g[:32] == GOLD["synth_gain_head"]).all() and crc(g) == int(GOLD["synth_gain_crc"])


def test_synth_statistics(orc):
    w = orc.synth_f16("layers.3.attention.wk.weight", orc.TENSOR_SEED, 0.02, 1 << 20).view(np.float16).astype(np.float64)
    assert abs(w.mean()) < 1e-4 and abs(w.std() - 0.02) < 2e-4 and np.abs(w).max() < 0.07
    a = orc.synth_f16("a", 1, 0.02, 1000); b = orc.synth_f16("b", 1, 0.02, 1000); c = orc.synth_f16("a", 2, 0.02, 1000)
    assert (a != b).any() and (a != c).any()
    assert (orc.synth_f16("a", 1, 0.02, 1000) == a).all()
    # prefix property: element i does not depend on n
    assert (orc.synth_f16("a", 1, 0.02, 10) == a[:10]).all()


# ------------------------------------------------------------------ per-kernel restatements
def test_k1_matvec_golden_and_float64(orc):
    a, W = GOLD["k1_a"], GOLD["k1_W"]
    out = orc.vector_mat_mul_trans(a, W, True)
    assert (out == GOLD["k1_out"]).all()
    ref = W.view(np.float16).astype(np.float64) @ a.astype(np.float64)
    assert np.abs(out - ref).max() < 1e-5
    assert np.abs(orc.vector_mat_mul_trans(a, W, False) - ref).max() < 1e-5


def test_k1_rejects_bad_shapes(orc):
    with pytest.raises(ValueError):
        orc.vector_mat_mul_trans(np.zeros(300, np.float32), np.zeros((2, 300), np.uint16), True)


def test_k4_k5_norm_golden_and_float64(orc):
    x = GOLD["k4_in"]
    out = orc.rms_norm(x)
    assert (out == GOLD["k4_out"]).all()
    ref = x.astype(np.float64) / np.sqrt((x.astype(np.float64) ** 2).mean(axis=1, keepdims=True) + 1e-6)
    assert np.abs(out - ref).max() < 1e-5
    assert (orc.row_element_multiply(x, GOLD["k5_gain"]) == GOLD["k5_out"]).all()


def test_k6_rope_golden_and_float64(orc):
    r = GOLD["k6_in"]
    out = orc.rope(r, 5)
    assert (out == GOLD["k6_out_past5"]).all()
    n_tok, H, D = r.shape
    ref = r.astype(np.float64).copy()
    j = np.arange(0, D, 2)
    for t in range(n_tok):
        ang = (5 + t) * 10000.0 ** (-j / D)
        x0, x1 = r[t, :, 0::2].astype(np.float64), r[t, :, 1::2].astype(np.float64)
        ref[t, :, 0::2] = x0 * np.cos(ang) - x1 * np.sin(ang)
        ref[t, :, 1::2] = x0 * np.sin(ang) + x1 * np.cos(ang)
    assert np.abs(out - ref).max() < 2e-5
    # position 0 is the identity; norms are preserved
    assert (orc.rope(r[:1], 0) == r[:1]).all()
    assert np.allclose(np.linalg.norm(out, axis=-1), np.linalg.norm(r, axis=-1), rtol=1e-5)


def test_k8_transpose(orc):
    a = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4)
    assert (orc.transpose_zy(a) == a.transpose(1, 0, 2)).all()


def test_k9_matmul_golden_and_float64(orc):
    out = orc.mat_mul(GOLD["k9_A"], GOLD["k9_Bt"], True, 0.125)
    assert (out == GOLD["k9_out_t"]).all()
    ref = np.einsum("bmk,bnk->bmn", GOLD["k9_A"].astype(np.float64), GOLD["k9_Bt"].astype(np.float64)) * 0.125
    assert np.abs(out - ref).max() < 1e-5
    out = orc.mat_mul(GOLD["k9_P"], GOLD["k9_V"], False, 1.0)
    assert (out == GOLD["k9_out_n"]).all()
    ref = np.einsum("bmk,bkn->bmn", GOLD["k9_P"].astype(np.float64), GOLD["k9_V"].astype(np.float64))
    assert np.abs(out - ref).max() < 1e-5


@pytest.mark.parametrize("T", [1, 5, 300])
def test_k10_softmax_golden(orc, T):
    s = GOLD[f"k10_in_T{T}"]
    out = orc.row_softmax(s)
    assert (out == GOLD[f"k10_out_T{T}"]).all()
    e = np.exp(s.astype(np.float64) - s.max(axis=1, keepdims=True))
    assert np.abs(out - e / e.sum(axis=1, keepdims=True)).max() < 1e-6
    assert np.allclose(out.sum(axis=1), 1.0, atol=1e-6)


def test_k11_k12_k13_elementwise(orc):
    u = GOLD["k12_in"]
    assert (orc.silu(u) == GOLD["k12_out"]).all()
    assert np.abs(orc.silu(u) - u.astype(np.float64) / (1 + np.exp(-u.astype(np.float64)))).max() < 1e-6
    assert (orc.addition(u, u) == u + u).all()
    assert (orc.element_mult(u, u) == u * u).all()


def test_q1_index_set(orc):
    """Defect Q1: cmdbuf_vector_reduce(...,8) skips r mod 4000 >= 3840 at V=32000 (1,280 logits)."""
    sk = orc.q1_skipped_indices(32000)
    assert len(sk) == 1280
    assert (sk == GOLD["q1_skipped_V32000"]).all()
    assert ((sk % 4000) >= 3840).all()
    assert len(orc.q1_skipped_indices(2048)) == 0


def test_lmhead_modes(orc):
    rng = np.random.default_rng(1)
    V, E = 32000, 512
    W = (rng.standard_normal((V, E)) * 0.05).astype(np.float16).view(np.uint16)
    x = rng.standard_normal(E).astype(np.float32)
    full = orc.lmhead(x, W, False)
    faith = orc.lmhead(x, W, True)
    Wf = W.view(np.float16).astype(np.float64)
    assert np.abs(full - Wf @ x).max() < 1e-4
    sk = orc.q1_skipped_indices(V)
    keep = np.setdiff1d(np.arange(V), sk)
    assert (full[keep] == faith[keep]).all()
    assert np.abs(faith[sk] - Wf[sk, : E // 2] @ x[: E // 2]).max() < 1e-4


def test_greedy_first_max_wins(orc):
    assert orc.greedy(np.array([0.0, 3.0, 3.0, 1.0], np.float32)) == 1
    assert orc.greedy(np.array([5.0, 3.0, 5.0], np.float32)) == 0
    assert orc.greedy(np.array([-2.0, -1.0], np.float32)) == 1


# ------------------------------------------------------------------ whole model
def _tensors(orc, shape):
    t = {}
    for name, dt, shp in shape.tensor_specs():
        n = int(np.prod(shp))
        t[name] = (orc.synth_f16(name, orc.TENSOR_SEED, orc.TENSOR_SIGMA, n).reshape(shp) if dt == "f16"
                   else orc.synth_gain(name, orc.TENSOR_SEED, orc.TENSOR_SIGMA, n))
    return t


@pytest.mark.parametrize("key", ["p1", "p2", "p17"])
def test_tiny_model_matches_golden_and_float64(orc, key):
    toks = GOLD["tiny_prompt_" + key].tolist()
    m = orc.OracleModel(orc.TINY); m.fill_synthetic()
    fast = orc.OracleModel(orc.TINY); fast.fill_synthetic()
    ref = RefModel(orc.TINY, _tensors(orc, orc.TINY))
    for i, t in enumerate(toks):
        lg, _ = m.eval(t, i, flags=orc.FAITHFUL_ORDER)
        lf, _ = fast.eval(t, i, flags=0)
        lr = ref.eval(t, i)
    assert (lg == GOLD["tiny_logits_" + key]).all()            # regression pin
    assert np.abs(lg - lr).max() < 2e-5                         # independent float64 forward
    assert np.abs(lf - lr).max() < 2e-5                         # fast flavour (CPU baseline) agrees
    assert orc.greedy(lg) == int(np.argmax(lr))


def test_tinyq1_model_modes_match_golden_and_float64(orc):
    toks = GOLD["tinyq1_prompt"].tolist()
    ref = RefModel(orc.TINY_Q1, _tensors(orc, orc.TINY_Q1))
    for mode, flag in (("correct", 0), ("faithful", orc.LM_FAITHFUL)):
        m = orc.OracleModel(orc.TINY_Q1); m.fill_synthetic()
        for i, t in enumerate(toks):
            lg, _ = m.eval(t, i, flags=orc.FAITHFUL_ORDER | flag)
        assert (lg == GOLD["tinyq1_logits_" + mode]).all()
    ref2 = RefModel(orc.TINY_Q1, _tensors(orc, orc.TINY_Q1))
    for i, t in enumerate(toks):
        lc = ref.eval(t, i, q1_faithful=False)
        lq = ref2.eval(t, i, q1_faithful=True)
    assert np.abs(GOLD["tinyq1_logits_correct"] - lc).max() < 2e-5
    assert np.abs(GOLD["tinyq1_logits_faithful"] - lq).max() < 2e-5
    sk = GOLD["q1_skipped_V32000"]
    assert (GOLD["tinyq1_logits_correct"][sk] != GOLD["tinyq1_logits_faithful"][sk]).mean() > 0.99


def test_layer_range_split_equals_full_model(orc):
    """Pipeline-stage semantics of the oracle: layers [0,1) then [1,2) == [0,2)."""
    full = orc.OracleModel(orc.TINY); full.fill_synthetic()
    a = orc.OracleModel(orc.TINY); a.fill_synthetic()
    b = orc.OracleModel(orc.TINY); b.fill_synthetic()
    for i, t in enumerate([1, 9, 300]):
        lg, _ = full.eval(t, i)
        _, h = a.eval(t, i, l0=0, l1=1, want_logits=False)
        lg2, _ = b.eval(None, i, l0=1, l1=2, hidden=h)
        assert (lg == lg2).all()


def test_kv_cache_layout(orc):
    """K/V rows land at [n_past, H, D] (th-llama.cpp:332-339) and unused rows stay zero."""
    m = orc.OracleModel(orc.TINY); m.fill_synthetic()
    m.eval(5, 0); m.eval(6, 1)
    k = m.kv(0, 0, 0)
    assert np.abs(k[0]).sum() > 0 and np.abs(k[1]).sum() > 0 and np.abs(k[2:]).sum() == 0
    m.reset_kv(0)
    assert np.abs(m.kv(0, 0, 0)).sum() == 0


# ------------------------------------------------------------------ A17 pinned by the reference's own functions
REF_FP16 = os.path.join(os.path.dirname(__file__), "golden", "ref_fp16.npz")


def test_fp16_converters_match_reference_fixture(orc):
    """tests/golden/ref_fp16.npz holds the outputs of the REFERENCE's ggml_compute_fp16_to_fp32 / fp32_to_fp16
    (th.cpp:294-359, compiled from /root/reference by `make -C oracle _ref`, generator tools/make_ref_fp16_golden.py):
    every binary16 pattern widened, and ~45k f32 inputs narrowed.  The oracle's restatement must reproduce them bit for bit."""
    g = np.load(REF_FP16)
    h = np.arange(65536, dtype=np.uint16)
    got = orc.fp16_to_fp32(h).view(np.uint32)
    ref = g["h2f_bits"]
    nan = ((h & 0x7C00) == 0x7C00) & ((h & 0x3FF) != 0)
    assert (got[~nan] == ref[~nan]).all()                               # every non-NaN pattern: identical bits (inf and denormals included)
    assert np.isnan(got[nan].view(np.float32)).all() and np.isnan(ref[nan].view(np.float32)).all()
    f = g["f_in_bits"].view(np.float32)
    assert (orc.fp32_to_fp16(f) == g["f2h"]).all()


def test_reference_fp16_library_when_present(orc):
    """In the build container (where /root/reference exists) the compiled reference functions are called directly."""
    import ctypes as C
    path = os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "_ref", "libth_ref_fp16.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built here (no /root/reference)")
    lib = C.CDLL(path)
    lib.ggml_compute_fp16_to_fp32.restype = C.c_float; lib.ggml_compute_fp16_to_fp32.argtypes = [C.c_uint16]
    lib.ggml_compute_fp32_to_fp16.restype = C.c_uint16; lib.ggml_compute_fp32_to_fp16.argtypes = [C.c_float]
    hs = np.arange(0, 65536, 37, dtype=np.uint16)
    mine = orc.fp16_to_fp32(hs)
    for h, v in zip(hs.tolist(), mine.tolist()):
        r = lib.ggml_compute_fp16_to_fp32(h)
        assert (np.isnan(r) and np.isnan(v)) or np.float32(r).view(np.uint32) == np.float32(v).view(np.uint32)
    xs = (np.random.default_rng(9).standard_normal(4000) * 50).astype(np.float32)
    assert [lib.ggml_compute_fp32_to_fp16(float(x)) for x in xs] == orc.fp32_to_fp16(xs).tolist()
