"""GPU parity tests, operator level: every thk_* operator (one per cmdbuf_* of th.hpp)
is called through the C-ABI and compared with the CPU oracle on the same seeded inputs.
Tolerance: north_star's 1e-3 on logits; operators are held to 2e-4 absolute on O(1)
values (f32 accumulation in a different order than the reference's strip+tree)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 2e-4


def rnd(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def f16w(rng, R, C, scale=0.02):
    return (rng.standard_normal((R, C)) * scale).astype(np.float16).view(np.uint16)


# (R, C): tiny/ragged rows, generic slot counts, and every compile-time specialised
# column class of the 7B/13B models (4096, 5120, 11008 half-slot, 13824)
MATVEC_SHAPES = [(8, 256), (5, 512), (64, 768), (33, 1536), (100, 2048), (4096, 4096), (11008, 4096), (4096, 11008),
                 (1024, 5120), (1024, 13824), (7, 11008), (2, 8192)]


@pytest.mark.parametrize("R,C", MATVEC_SHAPES)
def test_matvec_f16(ctx, orc, R, C):
    rng = np.random.default_rng(R * 131 + C)
    W, x = f16w(rng, R, C), rnd(rng, C)
    dW, dx, dy = ctx.from_numpy(W), ctx.from_numpy(x), ctx.alloc(R * 4)
    ctx.matvec_f16(dW, R, C, dx, dy)
    got = dy.download(np.float32, R)
    exp = orc.vector_mat_mul_trans(x, W, faithful=(R * C <= 1 << 22))
    assert np.abs(got - exp).max() < TOL * max(1.0, np.abs(exp).max())


@pytest.mark.parametrize("variant", [0, 1, 2, 5, 6])
@pytest.mark.parametrize("bpc", [1, 4, 8])
def test_matvec_variants_and_grids(ctx, orc, variant, bpc):
    """Every (rows/iteration, slots/batch) variant - batch loops 0-2, software-pipelined loops 5-6 - and grid size gives the
    same answer (a wave takes 1, 2 or more row groups depending on the grid, so the pipelined refill and its last-group
    epilogue are both exercised)."""
    rng = np.random.default_rng(variant * 7 + bpc)
    old = {k: ctx.get_tunable(k) for k in ("gemv_variant_wo", "gemv_variant_w2", "gemv_blocks_per_cu")}
    try:
        ctx.set_tunable("gemv_variant_wo", variant); ctx.set_tunable("gemv_variant_w2", variant)
        ctx.set_tunable("gemv_blocks_per_cu", bpc)
        for R, C in [(4096, 4096), (1023, 4096), (777, 11008), (515, 5120), (129, 13824), (300, 1024)]:
            W, x = f16w(rng, R, C), rnd(rng, C)
            dW, dx, dy = ctx.from_numpy(W), ctx.from_numpy(x), ctx.alloc(R * 4)
            ctx.matvec_f16(dW, R, C, dx, dy)
            got = dy.download(np.float32, R)
            exp = orc.vector_mat_mul_trans(x, W, faithful=False)
            assert np.abs(got - exp).max() < TOL * max(1.0, np.abs(exp).max()), (R, C)
    finally:
        for k, v in old.items():
            ctx.set_tunable(k, v)


def test_matvec_linearity_full_size(ctx):
    """Size-independent property at 7B size: W(ax+by) == a Wx + b Wy (no oracle needed)."""
    rng = np.random.default_rng(3)
    R, C = 11008, 4096
    dW = ctx.alloc(R * C * 2)
    ctx.synth_f16("layers.0.feed_forward.w1.weight", R * C, dW)
    x, y = rnd(rng, C), rnd(rng, C)
    outs = []
    for v in (x, y, (2.0 * x - 0.5 * y).astype(np.float32)):
        dv, do = ctx.from_numpy(v), ctx.alloc(R * 4)
        ctx.matvec_f16(dW, R, C, dv, do)
        outs.append(do.download(np.float32, R))
    assert np.abs(outs[2] - (2.0 * outs[0] - 0.5 * outs[1])).max() < 1e-4


def test_matvec_rejects_bad_shape(ctx, thk):
    d = ctx.alloc(4096)
    with pytest.raises(thk.ThkError, match="multiple of 256"):
        ctx.matvec_f16(d, 2, 300, d, d)


@pytest.mark.parametrize("rows,N", [(1, 256), (1, 4096), (3, 512), (8, 5120)])
def test_rms_norm_and_gain(ctx, orc, rows, N):
    rng = np.random.default_rng(N + rows)
    x, g = rnd(rng, rows, N, scale=3.0), (1 + 0.1 * rng.standard_normal(N)).astype(np.float32)
    dx, dg = ctx.from_numpy(x), ctx.from_numpy(g)
    ctx.rms_norm(dx, rows, N)
    got = dx.download(np.float32, (rows, N))
    exp = orc.rms_norm(x)
    assert np.abs(got - exp).max() < 1e-5
    ctx.row_element_multiply(dx, dg, rows, N)
    assert np.abs(dx.download(np.float32, (rows, N)) - orc.row_element_multiply(exp, g)).max() < 1e-5


@pytest.mark.parametrize("n_tok,H,D,n_past", [(1, 32, 128, 0), (1, 32, 128, 511), (8, 8, 64, 5), (1, 40, 128, 100)])
def test_rope(ctx, orc, n_tok, H, D, n_past):
    rng = np.random.default_rng(n_past + D)
    x = rnd(rng, n_tok, H, D)
    dx = ctx.from_numpy(x)
    ctx.rope(dx, n_tok, H, D, n_past)
    got = dx.download(np.float32, x.shape)
    assert np.abs(got - orc.rope(x, n_past)).max() < 1e-5


@pytest.mark.parametrize("rows,N", [(1, 1), (4, 5), (32, 300), (32, 512), (2, 1000)])
def test_row_softmax(ctx, orc, rows, N):
    rng = np.random.default_rng(rows * N)
    x = rnd(rng, rows, N, scale=3.0)
    dx = ctx.from_numpy(x)
    ctx.row_softmax(dx, rows, N)
    got = dx.download(np.float32, x.shape)
    assert np.abs(got - orc.row_softmax(x)).max() < 1e-6


def test_elementwise(ctx, orc):
    rng = np.random.default_rng(9)
    a, b = rnd(rng, 11008, scale=2.0), rnd(rng, 11008)
    da, db, dc = ctx.from_numpy(a), ctx.from_numpy(b), ctx.alloc(a.nbytes)
    ctx.add(da, db, dc, a.size)
    assert (dc.download(np.float32, a.size) == a + b).all()
    ctx.silu(da, a.size)
    s = da.download(np.float32, a.size)
    assert np.abs(s - orc.silu(a)).max() < 1e-6
    ctx.mul_inplace(da, db, a.size)
    assert np.abs(da.download(np.float32, a.size) - orc.element_mult(orc.silu(a), b)).max() < 1e-6


def oracle_attention(orc, q, Kc, Vc, T, H, D):
    """Reference chain: transposes + K9 (scale after sum) + K10 + K9 (th-llama.cpp:341-397)."""
    qT = orc.transpose_zy(q.reshape(1, H, D))
    Kw, Vw = orc.transpose_zy(Kc[:T].reshape(T, H, D)), orc.transpose_zy(Vc[:T].reshape(T, H, D))
    S = orc.mat_mul(qT, Kw, True, 1.0 / np.sqrt(np.float32(D)))
    P = orc.row_softmax(S.reshape(H, T)).reshape(H, 1, T)
    return orc.mat_mul(P, Vw, False, 1.0).reshape(H * D)


_ATTN_SHAPES = [(1, 32, 128, 512), (2, 32, 128, 512), (17, 8, 64, 64), (64, 8, 64, 64), (129, 32, 128, 512), (512, 32, 128, 512), (300, 40, 128, 512),
                (50, 12, 64, 64), (33, 4, 256, 64), (700, 32, 128, 2048)]
# (splits follow the live context | the cache capacity) x every shape, and - round 6 - 16 waves per workgroup on the D = 128 shapes (+ a 2048-position one: a
# 256-position split is exactly one round of 16 waves)
_ATTN_CASES = [(g, sh) for g in ((1,), (0,)) for sh in _ATTN_SHAPES] + [((1, 16), sh) for sh in _ATTN_SHAPES + [(2048, 32, 128, 2048)] if sh[2] == 128]


@pytest.mark.parametrize("splits", [1, 2, 4, 8])
@pytest.mark.parametrize("geom,shape", _ATTN_CASES, ids=[("dyn", "static")[1 - g[0]] + ("-16-waves" if len(g) > 1 else "") + "-%d-%d-%d-%d" % sh for g, sh in _ATTN_CASES])
def test_attn_decode(ctx, orc, splits, geom, shape):
    """K8+K9+K10+K9+K8 in one launch, in every launch geometry: splits follow the live context T | the cache capacity.  H = 12 / H = 4
    make H * splits a non-multiple of 8, D = 256 is the widest head, T = 700 of 2048 leaves most of the capacity unused (static
    splits: one workgroup per head gets everything)."""
    T, H, D, n_ctx = shape
    rng = np.random.default_rng(T * 31 + H)
    E = H * D
    q, Kc, Vc = rnd(rng, E), rnd(rng, n_ctx, E), rnd(rng, n_ctx, E)
    Kc[T:] = 1e6; Vc[T:] = 1e6   # rows beyond T must never be read
    names = ("attn_splits", "attn_tc_dyn", "attn_waves")[:1 + len(geom)]
    old = {k: ctx.get_tunable(k) for k in names}
    try:
        for k, v in zip(names, (splits,) + geom):
            ctx.set_tunable(k, v)
        dq, dk, dv, do = ctx.from_numpy(q), ctx.from_numpy(Kc), ctx.from_numpy(Vc), ctx.alloc(E * 4)
        ctx.attn_decode(dq, dk, dv, T, H, D, do)
        got = do.download(np.float32, E)
    finally:
        for k, v in old.items():
            ctx.set_tunable(k, v)
    exp = oracle_attention(orc, q, Kc, Vc, T, H, D)
    assert np.abs(got - exp).max() < 2e-5


@pytest.mark.parametrize("M,n_past,H,D,n_ctx", [(1, 0, 8, 64, 64), (5, 0, 8, 64, 64), (33, 7, 8, 64, 64), (128, 0, 32, 128, 512), (128, 384, 32, 128, 512)])
def test_attn_prefill_is_causal_decode_per_query(ctx, orc, M, n_past, H, D, n_ctx):
    """thk_attn_prefill: query i at position n_past+i == thk_attn_decode semantics with T = n_past+i+1 (oracle)."""
    rng = np.random.default_rng(M * 7 + n_past)
    E = H * D
    Q, Kc, Vc = rnd(rng, M, E), rnd(rng, n_ctx, E), rnd(rng, n_ctx, E)
    Kc[n_past + M:] = 1e6; Vc[n_past + M:] = 1e6   # rows beyond the last query's position must never be read
    dq, dk, dv, do = ctx.from_numpy(Q), ctx.from_numpy(Kc), ctx.from_numpy(Vc), ctx.alloc(M * E * 4)
    ctx.attn_prefill(dq, dk, dv, n_past, M, H, D, do)
    got = do.download(np.float32, (M, E))
    for i in sorted({0, M // 2, M - 1}):
        exp = oracle_attention(orc, Q[i], Kc, Vc, n_past + i + 1, H, D)
        assert np.abs(got[i] - exp).max() < 2e-5, i


def test_kv_append(ctx):
    rng = np.random.default_rng(5)
    H, D, n_ctx = 8, 64, 16
    k, v = rnd(rng, H * D), rnd(rng, H * D)
    dk, dv = ctx.from_numpy(k), ctx.from_numpy(v)
    kc, vc = ctx.alloc(n_ctx * H * D * 4), ctx.alloc(n_ctx * H * D * 4)
    ctx.kv_append(kc, vc, dk, dv, 3, H, D)
    K, V = kc.download(np.float32, (n_ctx, H * D)), vc.download(np.float32, (n_ctx, H * D))
    assert (K[3] == k).all() and (V[3] == v).all() and K[:3].sum() == 0 and K[4:].sum() == 0


@pytest.mark.parametrize("V,E", [(2048, 512), (32000, 512), (32000, 4096), (1000, 1024)])
@pytest.mark.parametrize("mode", [0, 1])
def test_lmhead_and_argmax(ctx, orc, V, E, mode):
    rng = np.random.default_rng(V + E + mode)
    W, x = f16w(rng, V, E, 0.05), rnd(rng, E)
    dW, dx, dl, did = ctx.from_numpy(W), ctx.from_numpy(x), ctx.alloc(V * 4), ctx.alloc(4)
    ctx.lmhead_f16(dW, V, E, dx, dl, mode)
    got = dl.download(np.float32, V)
    if V * E <= 1 << 24:
        exp = orc.lmhead(x, W, lm_faithful=bool(mode))
    else:   # big case: fast oracle + Q1 mask applied by hand
        Wf = W.view(np.float16).astype(np.float32)
        exp = Wf @ x
        if mode:
            sk = orc.q1_skipped_indices(V)
            exp[sk] = Wf[sk, : E // 2] @ x[: E // 2]
    assert np.abs(got - exp).max() < TOL * max(1.0, np.abs(exp).max())
    ctx.argmax(dl, V, did)
    assert int(did.download(np.int32, 1)[0]) == orc.greedy(got)


def test_argmax_first_max_wins(ctx):
    v = np.zeros(32000, np.float32); v[[31999, 777, 20000]] = 5.0; v[3] = -7.0
    dl, did = ctx.from_numpy(v), ctx.alloc(4)
    ctx.argmax(dl, v.size, did)
    assert int(did.download(np.int32, 1)[0]) == 777
    v = -np.ones(100, np.float32); v[42] = -0.5
    dl = ctx.from_numpy(v)
    ctx.argmax(dl, v.size, did)
    assert int(did.download(np.int32, 1)[0]) == 42


@pytest.mark.parametrize("V,k", [(32000, 41), (32000, 1), (32000, 1024), (2048, 40), (777, 777), (5, 3)])
def test_topk_f32(ctx, V, k):
    """Device top-k (radix select + bitonic sort in one workgroup) == numpy: the k largest values, descending, ties by ascending
    index - with heavy ties (quantised logits), negative values, signed zeros and infinities in the input."""
    rng = np.random.default_rng(V * 31 + k)
    lg = (rng.standard_normal(V) * 4).astype(np.float32)
    lg[::3] = np.round(lg[::3])                             # many exact ties
    if V > 100:
        lg[7] = np.inf; lg[11] = -np.inf; lg[13] = 0.0; lg[17] = -0.0; lg[19] = lg[23] = lg.max()
    d = ctx.from_numpy(lg)
    vals, ids = ctx.topk_f32(d, V, k)
    # reference order: value descending, then index ascending; -0.0 sorts below +0.0 in the kernel's bit-order map
    key = np.where(np.signbit(lg), ~lg.view(np.uint32), lg.view(np.uint32) | np.uint32(0x80000000)).astype(np.uint64)
    order = np.lexsort((np.arange(V), -key.astype(np.int64)))[:k]
    assert ids.tolist() == order.tolist()
    assert vals.view(np.uint32).tolist() == lg[order].view(np.uint32).tolist()


def test_topk_rejects_out_of_range(ctx, thk):
    d = ctx.alloc(40000 * 4)
    for V, k in [(40000, 10), (1000, 1025), (10, 11), (0, 1)]:
        with pytest.raises(thk.ThkError):
            ctx.topk_f32(d, V, k)


def test_embed(ctx, orc):
    rng = np.random.default_rng(2)
    V, E = 100, 512
    tab = f16w(rng, V, E, 1.0)
    dt, dx = ctx.from_numpy(tab), ctx.alloc(E * 4)
    ctx.embed_f16(dt, E, 37, dx)
    assert (dx.download(np.float32, E) == orc.fp16_to_fp32(tab[37])).all()


def test_synth_generator_bit_exact(ctx, orc):
    """The device-side synthetic-weight generator reproduces the oracle's bits."""
    n = 1 << 18
    d = ctx.alloc(n * 2)
    for name in ("tok_embeddings.weight", "layers.7.feed_forward.w3.weight"):
        ctx.synth_f16(name, n, d)
        assert (d.download(np.uint16, n) == orc.synth_f16(name, orc.TENSOR_SEED, orc.TENSOR_SIGMA, n)).all()
    g = ctx.alloc(4096 * 4)
    ctx.synth_gain_f32("norm.weight", 4096, g)
    assert (g.download(np.float32, 4096) == orc.synth_gain("norm.weight", orc.TENSOR_SEED, orc.TENSOR_SIGMA, 4096)).all()


@pytest.mark.parametrize("M,R,C", [(1, 64, 512), (8, 96, 512), (33, 128, 1024), (128, 4096, 4096), (128, 512, 11008), (200, 256, 512)])
def test_gemm_f16_prefill(ctx, orc, M, R, C):
    """MFMA prefill GEMM == M independent mat-vecs (prefill parity is defined against
    token-by-token decode)."""
    rng = np.random.default_rng(M + R + C)
    W, X = f16w(rng, R, C), rnd(rng, M, C)
    dW, dX, dY = ctx.from_numpy(W), ctx.from_numpy(X), ctx.alloc(M * R * 4)
    ctx.gemm_f16_prefill(dW, R, C, dX, M, dY)
    got = dY.download(np.float32, (M, R))
    exp = X.astype(np.float64) @ W.view(np.float16).astype(np.float64).T
    assert np.abs(got - exp).max() < TOL * max(1.0, np.abs(exp).max())
    row = orc.vector_mat_mul_trans(X[M - 1], W, faithful=False)
    assert np.abs(got[M - 1] - row).max() < TOL * max(1.0, np.abs(row).max())
