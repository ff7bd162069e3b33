"""GPU parity tests, model level: th_eval_gpu semantics through thk_model_* vs the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.npz"))
LOGIT_TOL = 1e-3   # north_star: logits within 1e-3


class fast_oracle:
    """`with fast_oracle(orc, wide) as flags:` - model-width oracle walks use the oracle's FAST flavour (AVX2 + OpenMP on all usable cores, flags = 0: the flavour the
    full-depth tests and bench.py's parity_check use; 511 tokens through two 4096-wide layers take 50 s in the reference-order scalar flavour, 2 s in this one);
    tiny widths keep the reference summation order (FAITHFUL_ORDER)."""

    def __init__(self, orc, wide):
        self.orc, self.wide = orc, wide

    def __enter__(self):
        self.threads = self.orc.num_threads()
        if self.wide:
            self.orc.set_num_threads(self.orc.usable_cpus())
        return 0 if self.wide else self.orc.FAITHFUL_ORDER

    def __exit__(self, *a):
        self.orc.set_num_threads(self.threads)
        return False


def make_pair(thk, orc, ctx, shape_name, n_seq=1, lm_mode=0, tunables=None, **kw):
    shape = getattr(thk, shape_name)
    old = {}
    for k, v in (tunables or {}).items():
        old[k] = ctx.get_tunable(k); ctx.set_tunable(k, v)
    try:
        m = thk.Model(ctx, shape, n_seq=n_seq, **kw)
        m.fill_synthetic()
        if lm_mode:
            m.set_lmhead_mode(lm_mode)
        m.finalize()
    finally:
        for k, v in old.items():
            ctx.set_tunable(k, v)
    om = orc.OracleModel(getattr(orc, shape_name), n_seq=n_seq)
    om.fill_synthetic()
    return m, om


@pytest.mark.parametrize("key", ["p1", "p2", "p17"])
def test_tiny_logits_match_golden_fixture(thk, orc, ctx, key):
    """HIP path vs the committed golden logits (prompts of 1, 2 and 17 tokens)."""
    m = thk.Model(ctx, thk.TINY); m.fill_synthetic(); m.finalize()
    toks = GOLD["tiny_prompt_" + key].tolist()
    for i, t in enumerate(toks):
        lg, _ = m.eval([t], i)
    assert np.abs(lg - GOLD["tiny_logits_" + key]).max() < LOGIT_TOL
    assert int(lg.argmax()) == int(GOLD["tiny_logits_" + key].argmax())
    m.close()


@pytest.mark.parametrize("splits", [1, 2, 4, 8])
@pytest.mark.parametrize("use_graph", [0, 1])
def test_tiny_model_every_token_vs_oracle(thk, orc, ctx, splits, use_graph):
    m, om = make_pair(thk, orc, ctx, "TINY", tunables={"attn_splits": splits, "use_graph": use_graph})
    rng = np.random.default_rng(splits)
    toks = [1] + rng.integers(3, 2048, 40).tolist()
    for i, t in enumerate(toks):
        lg, hid = m.eval([t], i, want_hidden=True)
        lo, ho = om.eval(t, i, flags=orc.FAITHFUL_ORDER)
        assert np.abs(lg - lo).max() < LOGIT_TOL, i
        assert np.abs(hid - ho).max() < LOGIT_TOL * max(1.0, np.abs(ho).max()), i
        assert int(lg.argmax()) == orc.greedy(lo)
    m.close()


R03_STEP = {"attn_tc_dyn": 0, "fold_finish": 0, "attn_splits": 4}     # the launch geometry of round 3's step


@pytest.mark.parametrize("tunables", [{"attn_waves": 4}, {"attn_waves": 4, "attn_splits": 8}, {"attn_waves": 8}, {"attn_waves": 8, "attn_splits": 8}, {"fold_embed": 0}, {"fold_embed": 0, "use_graph": 0},
                                      {"attn_tc_dyn": 0}, {"attn_tc_dyn": 0, "attn_waves": 4},
                                      {"fold_finish": 0}, {"fold_finish": 0, "use_graph": 0}, {"fold_finish": 1, "use_graph": 0}, R03_STEP,
                                      {"gemv_grid_qkv": 5, "gemv_grid_wo": 3, "gemv_grid_w13": 7, "gemv_grid_w2": 1, "gemv_grid_head": 11}])
def test_optional_paths_vs_oracle(thk, orc, ctx, tunables):
    """The off-by-default options stay correct: 4-wave attention blocks, the stand-alone embedding launch (default: the row is
    fetched by layer 0's qkv prologue), splits over the cache capacity (default: over the live context), the greedy pick as a launch of its own (default: folded into
    the lm-head launch's last workgroup), explicit workgroup counts for the mat-vecs (odd ones, fewer than one per CU)."""
    m, om = make_pair(thk, orc, ctx, "TINY", tunables=tunables)
    rng = np.random.default_rng(5)
    toks = [1] + rng.integers(3, 2048, 30).tolist()
    for i, t in enumerate(toks):
        lg, _ = m.eval([t], i); lo, _ = om.eval(t, i)
        assert np.abs(lg - lo).max() < LOGIT_TOL, i
    m.seq_set(0, 7, len(toks))
    for _ in range(6):
        m.decode_step(0, advance=True)
    gen, n, pos = m.seq_get(0)
    tok, exp = 7, []
    for i in range(6):
        lo, _ = om.eval(tok, len(toks) + i); tok = orc.greedy(lo); exp.append(tok)
    assert gen.tolist() == exp
    m.close()


def test_multi_token_eval_equals_one_by_one(thk, orc, ctx):
    """th_eval_gpu with n_tokens>1 == tokens fed one at a time (kAllowedSubsequentBatchSize=1)."""
    m, om = make_pair(thk, orc, ctx, "TINY")
    toks = [1, 50, 60, 70, 80]
    lg, _ = m.eval(toks, 0)
    for i, t in enumerate(toks):
        lo, _ = om.eval(t, i)
    assert np.abs(lg - lo).max() < LOGIT_TOL
    m.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_q1_lmhead_modes(thk, orc, ctx, mode):
    """faithful|correct lm-head switch at V=32000 where the defect is observable."""
    m, om = make_pair(thk, orc, ctx, "TINY_Q1", lm_mode=mode)
    toks = GOLD["tinyq1_prompt"].tolist()
    for i, t in enumerate(toks):
        lg, _ = m.eval([t], i)
        lo, _ = om.eval(t, i, flags=orc.FAITHFUL_ORDER | (orc.LM_FAITHFUL if mode else 0))
    assert np.abs(lg - lo).max() < LOGIT_TOL
    gold = GOLD["tinyq1_logits_" + ("faithful" if mode else "correct")]
    assert np.abs(lg - gold).max() < LOGIT_TOL
    other = GOLD["tinyq1_logits_" + ("correct" if mode else "faithful")]
    sk = GOLD["q1_skipped_V32000"]
    assert np.abs(lg[sk] - other[sk]).max() > 10 * LOGIT_TOL   # the two modes really differ there
    m.close()


def test_device_decode_loop_matches_eval_path(thk, orc, ctx):
    """Stream-ordered greedy loop (graph replay, no host round trip) == host-driven eval + oracle greedy."""
    m, om = make_pair(thk, orc, ctx, "TINY")
    n = 30
    m.seq_set(0, 1, 0)
    for _ in range(n):
        m.decode_step(0, advance=True)
    gen, ngen, pos = m.seq_get(0)
    assert ngen == n and pos == n
    tok, exp = 1, []
    for i in range(n):
        lo, _ = om.eval(tok, i)
        tok = orc.greedy(lo); exp.append(tok)
    assert gen.tolist() == exp
    m.close()


def test_multi_step_graph_matches_single_steps(thk, orc, ctx):
    """thk_model_decode_steps (8 steps per captured graph + remainder) == repeated thk_model_decode_step."""
    a, om = make_pair(thk, orc, ctx, "TINY")
    b = thk.Model(ctx, thk.TINY); b.fill_synthetic(); b.finalize()
    a.seq_set(0, 1, 0); b.seq_set(0, 1, 0)
    a.decode_steps(27, 0, advance=True)
    for _ in range(27):
        b.decode_step(0, advance=True)
    ga, na, pa = a.seq_get(0); gb, nb, pb = b.seq_get(0)
    assert na == nb == 27 and pa == pb == 27 and ga.tolist() == gb.tolist()
    a.decode_steps(3, 0, advance=True)           # fewer than one multi-step graph
    assert a.seq_get(0)[2] == 30
    a.close(); b.close()


def test_multi_step_graph_cache_is_bounded(thk, ctx):
    """A sequence keeps at most 6 multi-step graphs (least recently used evicted): many distinct step counts neither leak
    graphs nor change results."""
    a = thk.Model(ctx, thk.TINY); a.fill_synthetic(); a.finalize()
    b = thk.Model(ctx, thk.TINY); b.fill_synthetic(); b.finalize()
    a.seq_set(0, 1, 0); b.seq_set(0, 1, 0)
    total = 0
    for n in (2, 3, 4, 5, 6, 7, 8, 9, 3, 2, 9):          # 8 distinct counts > the cache's 6 slots; 3, 2 and 9 come round again
        a.decode_steps(n, 0, advance=True); total += n
    for _ in range(total):
        b.decode_step(0, advance=True)
    ga, na, pa = a.seq_get(0); gb, nb, pb = b.seq_get(0)
    assert na == nb == total and pa == pb == total and ga.tolist() == gb.tolist()
    a.close(); b.close()


def test_folded_greedy_pick_over_many_steps_and_sequences(thk, orc, ctx):
    """The lm-head launch's last workgroup finishes the token (ticket counter zeroed for the next launch): 60 consecutive steps on
    two interleaved sequences give the tokens of the stand-alone finish_token launch, in graph replay and eagerly."""
    out = {}
    for fold in (1, 0, 2):       # 2: workgroup 0 is the one that waits - dispatched FIRST, it spins through the whole launch: the pick does not lean on the dispatch order
        for graph in (1, 0):
            m, om = make_pair(thk, orc, ctx, "TINY", n_seq=2, tunables={"fold_finish": fold, "use_graph": graph})
            om.close()
            m.seq_set(0, 1, 0); m.seq_set(1, 9, 0)
            for k in range(6):
                m.decode_steps(5, 0, advance=True); m.decode_steps(5, 1, advance=True)
            out[(fold, graph)] = (m.seq_get(0)[0].tolist(), m.seq_get(1)[0].tolist())
            assert len(out[(fold, graph)][0]) == 30
            m.close()
    assert out[(1, 1)] == out[(0, 1)] == out[(1, 0)] == out[(0, 0)] == out[(2, 1)] == out[(2, 0)]


def test_folded_pick_by_the_first_dispatched_workgroup_at_model_width(thk, ctx):
    """fold_finish = 2 on the 7B-wide lm-head (V = 32000: 2048 workgroups, and with gemv_grid_head = 8192 four generations of them): workgroup 0 - on the device
    before any other - waits for every key.  Same tokens as the default and as the stand-alone pick, no time-out: forward progress needs no dispatch order."""
    shape = thk.ModelShape(n_layer=1)
    outs = {}
    for cfg in ({"fold_finish": 1}, {"fold_finish": 2}, {"fold_finish": 0}, {"fold_finish": 2, "gemv_grid_head": 8192}):
        for k, v in cfg.items():
            ctx.set_tunable(k, v)
        try:
            m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
        finally:
            ctx.set_tunable("fold_finish", 1); ctx.set_tunable("gemv_grid_head", 0)
        m.seq_set(0, 1, 0)
        m.decode_steps(40, 0, advance=True)
        outs[tuple(cfg.items())] = m.seq_get(0)[0].tolist()         # THK_ERR_STATE here if the in-launch wait had timed out
        m.close()
    vals = list(outs.values())
    assert all(v == vals[0] and len(v) == 40 for v in vals)


def test_device_step_clock(thk, ctx):
    """thk_model_seq_clock: one stamp of the chip-wide 100 MHz counter per logged step, strictly increasing, also inside
    replayed multi-step graphs; seq_set clears it together with the token log."""
    m = thk.Model(ctx, thk.TINY); m.fill_synthetic(); m.finalize()
    m.seq_set(0, 1, 0)
    m.decode_steps(40, 0, advance=True)
    gen, n, pos = m.seq_get(0)
    clk = m.seq_clock(0).astype(np.int64)
    assert n == 40 and clk.size == 40 and (np.diff(clk) > 0).all()
    assert 0 < np.diff(clk).mean() * 1e-2 < 5000            # a tiny-model step takes microseconds, not seconds (10 ns ticks)
    m.seq_set(0, 1, 0)
    m.decode_step(0, advance=True)
    assert m.seq_clock(0).size == 1
    m.close()


def test_hold_position_protocol(thk, orc, ctx):
    """advance=0 re-evaluates the same cache slot (fixed-T benchmark protocol): idempotent logits."""
    m, om = make_pair(thk, orc, ctx, "TINY")
    for i, t in enumerate([1, 9, 33]):
        m.eval([t], i); om.eval(t, i)
    m.seq_set(0, 77, 3)
    for _ in range(3):
        m.decode_step(0, advance=False)
    gen, ngen, pos = m.seq_get(0)
    lo, _ = om.eval(77, 3)
    t1 = orc.greedy(lo)
    lo2, _ = om.eval(t1, 3)
    t2 = orc.greedy(lo2)
    assert pos == 3 and ngen == 3 and gen[0] == t1 and gen[1] == t2
    m.close()


def test_multiple_sequences_are_independent(thk, orc, ctx):
    m, om = make_pair(thk, orc, ctx, "TINY", n_seq=3)
    prompts = [[1, 5, 6], [1, 900, 3, 4], [1]]
    for s, p in enumerate(prompts):
        for i, t in enumerate(p):
            lg, _ = m.eval([t], i, seq=s)
            lo, _ = om.eval(t, i, seq=s)
        assert np.abs(lg - lo).max() < LOGIT_TOL
    # interleave: another token on seq 0 after touching the others
    lg, _ = m.eval([42], 3, seq=0); lo, _ = om.eval(42, 3, seq=0)
    assert np.abs(lg - lo).max() < LOGIT_TOL
    m.reset_kv(1)
    lg, _ = m.eval([1], 0, seq=1); om.reset_kv(1); lo, _ = om.eval(1, 0, seq=1)
    assert np.abs(lg - lo).max() < LOGIT_TOL
    m.close()


def test_pipeline_stages_on_one_gpu(thk, orc, ctx):
    """Layer-range stages (config C4 semantics): [0,1) + [1,2) with the hidden hand-off == full model."""
    shape = thk.TINY
    full, om = make_pair(thk, orc, ctx, "TINY")
    a = thk.Model(ctx, shape, 0, 1, flags=thk.THK_STAGE_EMBED); a.fill_synthetic(); a.finalize()
    b = thk.Model(ctx, shape, 1, 2, flags=thk.THK_STAGE_HEAD); b.fill_synthetic(); b.finalize()
    for i, t in enumerate([1, 8, 99, 1000]):
        lg, _ = full.eval([t], i)
        _, h = a.eval([t], i, want_logits=False, want_hidden=True)
        lg2, _ = b.eval(None, i, hidden=h)
        assert np.abs(lg - lg2).max() < 1e-5, i
    for mm in (full, a, b):
        mm.close()


@pytest.mark.parametrize("E,H,cuts", [(512, 8, (0, 2, 4)), (512, 8, (0, 1, 2, 3, 4)), (4096, 32, (0, 1, 3, 4))], ids=["tiny-2-stages", "tiny-4-stages", "7B-width-3-stages"])
@pytest.mark.parametrize("M,n_past", [(300, 0), (130, 9)])
def test_prefill_stages_on_one_gpu(thk, orc, ctx, E, H, cuts, M, n_past):
    """thk_model_prefill_stage (round 6): the MFMA prompt pass on layer-range stages, the M x E rows handed on in ONE device buffer, against the
    full-model thk_model_prefill (same arithmetic per layer; a stage's first layer takes the power-of-two scale of its operand image from the row itself
    where the full model's layer takes it from the previous layer's ffn-norm input, so 'equal' is to rounding, not to the bit) and against the ORACLE
    fed token by token (1e-3), followed by two decode steps through the stages on the caches the pass filled.  Semantics: the reference's batch
    branch is per layer (th-llama.cpp:305-311, :365-404)."""
    L = cuts[-1]
    shape = thk.ModelShape(n_vocab=2048, n_embd=E, n_mult=256, n_head=H, n_layer=L, n_ctx=448)
    oshape = orc.ModelShape(n_vocab=2048, n_embd=E, n_mult=256, n_head=H, n_layer=L, n_ctx=448)
    full = thk.Model(ctx, shape); full.fill_synthetic(); full.finalize()
    stages = []
    for k in range(len(cuts) - 1):
        flags = (thk.THK_STAGE_EMBED if k == 0 else 0) | (thk.THK_STAGE_HEAD if k == len(cuts) - 2 else 0)
        st = thk.Model(ctx, shape, cuts[k], cuts[k + 1], flags=flags); st.fill_synthetic(); st.finalize()
        stages.append(st)
    om = orc.OracleModel(oshape); om.fill_synthetic()
    rng = np.random.default_rng(31 * M + n_past)
    toks = np.concatenate([[1], rng.integers(3, 2048, n_past + M + 2)]).astype(np.int32)
    rows = thk.Buffer(ctx, M * E * 4)

    def through_stages(tok, pos):
        h = None
        for st in stages:
            lg, h = st.eval([tok] if st is stages[0] else None, pos, hidden=h, want_logits=st is stages[-1], want_hidden=st is not stages[-1])
        return lg

    for i in range(n_past):                                 # rows a decode step wrote, on both sides
        full.eval([int(toks[i])], i, want_logits=False)
        through_stages(int(toks[i]), i)
    lf = full.prefill(toks[n_past:n_past + M], n_past)
    ls = None
    for st in stages:
        ls = st.prefill_stage(toks[n_past:n_past + M] if st is stages[0] else None, rows, M, n_past, want_logits=st is stages[-1])
    ctx.sync()
    lo = None
    with fast_oracle(orc, E > 512) as fl:
        for i in range(n_past + M):
            lo, _ = om.eval(int(toks[i]), i, want_logits=(i == n_past + M - 1), flags=fl)
        d_full, d_orc = float(np.abs(ls - lf).max()), float(np.abs(ls - lo).max())
        print(f"\n[stage prefill E={E} cuts={cuts} M={M} n_past={n_past}] vs full-model prefill {d_full:.3e}, vs oracle {d_orc:.3e}")
        assert d_full < 1e-4, d_full
        assert d_orc < LOGIT_TOL, d_orc
        assert int(ls.argmax()) == orc.greedy(lo)
        for i in (n_past + M, n_past + M + 1):                  # decode through the stages on the caches the stage passes filled
            lg = through_stages(int(toks[i]), i); lo, _ = om.eval(int(toks[i]), i, flags=fl)
            assert np.abs(lg - lo).max() < LOGIT_TOL, (i, float(np.abs(lg - lo).max()))
            assert int(lg.argmax()) == orc.greedy(lo)
    # argument errors: a stage without the table needs the rows, a stage without the head cannot give logits
    with pytest.raises(thk.ThkError):
        stages[-1].prefill_stage(None, None, M, n_past)
    with pytest.raises(thk.ThkError):
        stages[0].prefill_stage(toks[:4], rows, 4, 0, want_logits=True)
    with pytest.raises(thk.ThkError):
        stages[0].prefill(toks[:4], 0)                      # thk_model_prefill stays the full-model entry point
    rows.free(); om.close()
    for mm in [full] + stages:
        mm.close()


def test_loaded_tensors_equal_synthetic_fill(thk, orc, ctx):
    """thk_model_set_tensor (the loader's upload path) with host tensors == device-side synthetic fill."""
    shape = thk.TINY
    a = thk.Model(ctx, shape); a.fill_synthetic(); a.finalize()
    b = thk.Model(ctx, shape)
    for name, dt, shp in orc.TINY.tensor_specs():
        n = int(np.prod(shp))
        arr = orc.synth_f16(name, orc.TENSOR_SEED, orc.TENSOR_SIGMA, n).reshape(shp) if dt == "f16" else \
            orc.synth_gain(name, orc.TENSOR_SEED, orc.TENSOR_SIGMA, n)
        b.set_tensor(name, arr)
    b.finalize()
    for i, t in enumerate([1, 2, 3]):
        la, _ = a.eval([t], i); lb, _ = b.eval([t], i)
        assert (la == lb).all()
    with pytest.raises(thk.ThkError, match="unknown tensor"):
        b.set_tensor("layers.0.nope.weight", np.zeros(4, np.float32))
    with pytest.raises(thk.ThkError, match="shape"):
        b.set_tensor("norm.weight", np.zeros(7, np.float32))
    a.close(); b.close()


def test_error_paths(thk, ctx):
    m = thk.Model(ctx, thk.TINY)
    with pytest.raises(thk.ThkError, match="finalize"):
        m.eval([1], 0)
    m.fill_synthetic(); m.finalize()
    with pytest.raises(thk.ThkError, match="n_ctx"):
        m.eval([1], thk.TINY.n_ctx)
    with pytest.raises(thk.ThkError, match="out of range"):
        m.eval([thk.TINY.n_vocab], 0)
    with pytest.raises(thk.ThkError):
        thk.Model(ctx, thk.ModelShape(n_embd=500))
    m.close()


# ------------------------------------------------------------------ 7B / 13B dimensions
GEMV_KERNELS = ("qkv", "wo", "w13", "w2", "head")


@pytest.mark.parametrize("variant", [None, 0, 1, 2, 5, 6, 8])
@pytest.mark.parametrize("E,H,L,name", [(4096, 32, 2, "7B-dims"), (5120, 40, 1, "13B-dims")])
def test_full_width_layers_vs_oracle(thk, orc, ctx, E, H, L, name, variant):
    """Real 7B/13B row geometry (E, F=11008/13824, V=32000, T up to 512) on a 1-2 layer model: exercises every
    compile-time-specialised kernel against the oracle's fast flavour - with the default launch geometry (None) and with every
    mat-vec forced to one loop variant (batch loops 0-2, software-pipelined loops 5-6), so each fused prologue/epilogue
    (norm, embedding fetch, split combine | RoPE + K/V append, residual, SwiGLU, lm-head + arg-max) runs in both loop forms.
    Variants 1 and 6 are the single-row forms of qkv and w13 (the RoPE / SwiGLU pair meets in LDS), 8 the quarter-row
    form of w2 (a workgroup per row; the other kernels take variant 0 then)."""
    tun = {} if variant is None else {f"gemv_variant_{k}": variant for k in GEMV_KERNELS}
    old = {k: ctx.get_tunable(k) for k in tun}
    for k, v in tun.items():
        ctx.set_tunable(k, v)
    try:
        shape = thk.ModelShape(n_embd=E, n_head=H, n_layer=L)
        oshape = orc.ModelShape(n_embd=E, n_head=H, n_layer=L)
        m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
    finally:
        for k, v in old.items():
            ctx.set_tunable(k, v)
    om = orc.OracleModel(oshape); om.fill_synthetic()
    rng = np.random.default_rng(E)
    toks = [1] + rng.integers(3, 32000, 5).tolist()
    for i, t in enumerate(toks):
        lg, _ = m.eval([t], i); lo, _ = om.eval(t, i, flags=0)
        assert np.abs(lg - lo).max() < LOGIT_TOL, (name, i)
        assert int(lg.argmax()) == orc.greedy(lo)
    m.close(); om.close()


def test_single_row_forms_fall_back_when_a_wave_would_own_too_many_rows(thk, orc, ctx):
    """The single-row forms of qkv / w13 keep their pair sums in 32 LDS rounds per wave and the quarter-row w2 in 32 rows per
    workgroup: with a grid too small for that (8 workgroups here) finalize takes the row-pair / whole-row forms instead of failing
    at the first launch - same logits as the default geometry within the tolerance."""
    shape = thk.ModelShape(n_embd=4096, n_head=32, n_layer=1)
    oshape = orc.ModelShape(n_embd=4096, n_head=32, n_layer=1)
    tun = {"gemv_grid_qkv": 8, "gemv_grid_w13": 8, "gemv_grid_w2": 8}
    old = {k: ctx.get_tunable(k) for k in tun}
    for k, v in tun.items():
        ctx.set_tunable(k, v)
    try:
        m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
    finally:
        for k, v in old.items():
            ctx.set_tunable(k, v)
    om = orc.OracleModel(oshape); om.fill_synthetic()
    for i, t in enumerate([1, 77, 4242]):
        lg, _ = m.eval([t], i); lo, _ = om.eval(t, i, flags=0)
        assert np.abs(lg - lo).max() < LOGIT_TOL, i
        assert int(lg.argmax()) == orc.greedy(lo)
    m.close(); om.close()


def test_7b_full_model_properties(thk, ctx):
    """Full 7B (13.2 GB of synthetic f16 weights) at T=512 — size-independent properties:
    determinism, hold-position idempotence, device loop == eval path greedy tokens,
    two half-models chained == full model."""
    shape = thk.LLAMA_7B
    m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
    rng = np.random.default_rng(511)
    prompt = [1] + rng.integers(3, 32000, 15).tolist()
    lg1, h1 = m.eval(prompt, 0, want_hidden=True)
    assert np.isfinite(lg1).all() and np.isfinite(h1).all()
    m.reset_kv(0)
    lg2, _ = m.eval(prompt, 0)
    assert (lg1 == lg2).all()                                    # deterministic (no atomics in the data path)
    # greedy continuation: host-driven vs device loop
    toks_host, lg = [], lg1
    for i in range(4):
        t = int(lg.argmax()); toks_host.append(t)
        lg, _ = m.eval([t], len(prompt) + i)
    m.reset_kv(0)
    m.eval(prompt[:-1], 0, want_logits=False)
    m.seq_set(0, prompt[-1], len(prompt) - 1)
    for _ in range(4):
        m.decode_step(0, advance=True)
    gen, n, pos = m.seq_get(0)
    assert gen.tolist() == toks_host and pos == len(prompt) + 3
    # last cache slot (T=512): finite, idempotent
    m.seq_set(0, 5, 511)
    m.decode_step(0, advance=False); a, _, _ = m.seq_get(0)
    m.seq_set(0, 5, 511)
    m.decode_step(0, advance=False); b, _, _ = m.seq_get(0)
    assert a[0] == b[0]
    m.close()
    # two chained half-model stages == full model
    a = thk.Model(ctx, shape, 0, 16, flags=thk.THK_STAGE_EMBED); a.fill_synthetic(); a.finalize()
    b = thk.Model(ctx, shape, 16, 32, flags=thk.THK_STAGE_HEAD); b.fill_synthetic(); b.finalize()
    for i, t in enumerate(prompt):
        _, h = a.eval([t], i, want_logits=False, want_hidden=True)
        lgp, _ = b.eval(None, i, hidden=h)
    assert np.abs(lgp - lg1).max() < 1e-5
    a.close(); b.close()


# ------------------------------------------------------------------ batched prefill (config C3)
@pytest.mark.parametrize("M,n_past", [(1, 0), (5, 0), (17, 0), (33, 3), (56, 4)])
def test_prefill_equals_token_by_token(thk, orc, ctx, M, n_past):
    """MFMA prefill == feeding the tokens one at a time (prefill parity is defined against decode,
    SURVEY.md Q5) == the oracle; the KV rows it appends let decode continue seamlessly."""
    m, om = make_pair(thk, orc, ctx, "TINY")
    rng = np.random.default_rng(M)
    toks = [1] + rng.integers(3, 2048, n_past + M).tolist()
    for i in range(n_past):                      # context that already exists
        m.eval([toks[i]], i); om.eval(toks[i], i)
    lp = m.prefill(toks[n_past:n_past + M], n_past)
    for i in range(n_past, n_past + M):
        lo, _ = om.eval(toks[i], i)
    assert np.abs(lp - lo).max() < LOGIT_TOL
    assert int(lp.argmax()) == orc.greedy(lo)
    # decode continues from the prefilled cache
    nxt = toks[n_past + M]
    lg, _ = m.eval([nxt], n_past + M); lo2, _ = om.eval(nxt, n_past + M)
    assert np.abs(lg - lo2).max() < LOGIT_TOL
    m.close()


@pytest.mark.parametrize("M,n_past", [(129, 0), (200, 7), (256, 0)])
def test_prefill_in_slabs_of_128(thk, ctx, M, n_past):
    """Prompts longer than 128 tokens go through the layers in slabs; later slabs attend to the rows the earlier
    ones cached.  Compared with token-by-token decode of the same prompt on a second model instance."""
    shape = thk.ModelShape(n_vocab=2048, n_embd=512, n_mult=256, n_head=8, n_layer=2, n_ctx=320)
    a = thk.Model(ctx, shape); a.fill_synthetic(); a.finalize()
    b = thk.Model(ctx, shape); b.fill_synthetic(); b.finalize()
    rng = np.random.default_rng(M)
    toks = np.concatenate([[1], rng.integers(3, 2048, n_past + M)]).astype(np.int32)
    if n_past:
        a.eval(toks[:n_past], 0)
    lp = a.prefill(toks[n_past:n_past + M], n_past)
    ld, _ = b.eval(toks[:n_past + M], 0)
    assert np.abs(lp - ld).max() < LOGIT_TOL
    assert int(lp.argmax()) == int(ld.argmax())
    nxt = int(toks[n_past + M])
    la, _ = a.eval([nxt], n_past + M); lb, _ = b.eval([nxt], n_past + M)
    assert np.abs(la - lb).max() < LOGIT_TOL
    a.close(); b.close()


# tiny width: every slab edge; 7B / 13B width (round 6: the 4096 | 11008 and 5120 | 13824 column classes of gemm_prefill_v3h_kernel and the D = 128 slab
# attention, two layers): one full slab, slab + pad-tile slab behind decode-written rows, the 511-token prompt of bench.py's prompt_512_tokens_ms
_SLAB_EDGE_CASES = [(512, 8, M, n_past) for M, n_past in [(127, 0), (128, 1), (255, 0), (256, 30), (257, 0), (384, 17), (385, 0), (512, 5)]] \
    + [(4096, 32, 256, 0), (4096, 32, 257, 30), (4096, 32, 511, 0), (5120, 40, 256, 0), (5120, 40, 300, 9)]


@pytest.mark.parametrize("E,H,M,n_past", _SLAB_EDGE_CASES)
def test_prefill_slab_edges_vs_oracle(thk, orc, ctx, E, H, M, n_past):
    """Prompt lengths on both sides of every slab edge (one short of a slab, exactly one, one more; 256 + 128, 256 + 129, two full slabs), alone and behind
    rows a decode step wrote, against the ORACLE fed token by token: the last prompt position's logits, then two decode steps on the cache the slabs filled
    (a wrong K/V row, pad-tile leak or image offset at an edge would show there).  Semantics: th-llama.cpp:464-660 with the batch branch :305-311."""
    shape = thk.ModelShape(n_vocab=2048, n_embd=E, n_mult=256, n_head=H, n_layer=2, n_ctx=576)
    oshape = orc.ModelShape(n_vocab=2048, n_embd=E, n_mult=256, n_head=H, n_layer=2, n_ctx=576)
    m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
    om = orc.OracleModel(oshape); om.fill_synthetic()
    rng = np.random.default_rng(7 * M + n_past)
    toks = np.concatenate([[1], rng.integers(3, 2048, n_past + M + 2)]).astype(np.int32)
    if n_past:
        m.eval(toks[:n_past], 0, want_logits=False)
    lp = m.prefill(toks[n_past:n_past + M], n_past)
    lo = None
    with fast_oracle(orc, E > 512) as fl:
        for i in range(n_past + M):
            lo, _ = om.eval(int(toks[i]), i, want_logits=(i == n_past + M - 1), flags=fl)
        assert np.abs(lp - lo).max() < LOGIT_TOL, (M, n_past, float(np.abs(lp - lo).max()))
        assert int(lp.argmax()) == orc.greedy(lo)
        for i in (n_past + M, n_past + M + 1):
            lg, _ = m.eval([int(toks[i])], i); lo, _ = om.eval(int(toks[i]), i, flags=fl)
            assert np.abs(lg - lo).max() < LOGIT_TOL, (M, n_past, i)
            assert int(lg.argmax()) == orc.greedy(lo)
    m.close(); om.close()


@pytest.mark.parametrize("rms", [1.0, 60.0, 1000.0])
def test_prefill_deferred_norm_range(thk, ctx, rms):
    """The deferred norm carries a token's sum of squares as 2^-32 fixed point in 64 bits (thk_prefill.hip, ssq_fixed): rows up to a sum of squares of 2^32
    are inside its range - rms < 2896 at E = 512, < 1024 at E = 4096; LLaMA's residual stream with its outlier channels has rms of a few tens.  An embedding
    table scaled to rms 1, 60 and 1000: the 9-launch prompt pass, the explicit-norm pass (prefill_deferred_norm = 0) and token-by-token decode
    (RMSNorm in f32 inside the mat-vec prologue) agree on the logits."""
    shape = thk.ModelShape(n_vocab=2048, n_embd=512, n_mult=256, n_head=8, n_layer=2, n_ctx=320)
    rng = np.random.default_rng(int(rms))
    table = (rng.standard_normal((shape.n_vocab, shape.n_embd)) * rms).astype(np.float16)
    toks = np.concatenate([[1], rng.integers(3, 2048, 299)]).astype(np.int32)
    out = {}
    for name in ("deferred", "explicit", "decode"):
        m = thk.Model(ctx, shape); m.fill_synthetic(); m.set_tensor("tok_embeddings.weight", table); m.finalize()
        if name == "decode":
            out[name], _ = m.eval(toks, 0)
        else:
            ctx.set_tunable("prefill_deferred_norm", 1 if name == "deferred" else 0)
            try:
                out[name] = m.prefill(toks, 0)
            finally:
                ctx.set_tunable("prefill_deferred_norm", 1)
        m.close()
    assert np.isfinite(out["deferred"]).all()
    d1, d2 = float(np.abs(out["deferred"] - out["decode"]).max()), float(np.abs(out["explicit"] - out["decode"]).max())
    print(f"\n[deferred-norm range] residual rms {rms:g}: deferred vs decode {d1:.3e}, explicit vs decode {d2:.3e}")
    assert d1 < LOGIT_TOL and d2 < LOGIT_TOL
    assert int(out["deferred"].argmax()) == int(out["decode"].argmax())


def test_context_beyond_512(thk, orc, ctx):
    """n_ctx is a parameter, not the reference's compile-time 512 (th-llama.hpp:105): 1100 prompt tokens through the
    slab prefill, then decode steps at T > 1100, against the oracle fed token by token."""
    shape = thk.ModelShape(n_vocab=2048, n_embd=512, n_mult=256, n_head=8, n_layer=2, n_ctx=1200)
    oshape = orc.ModelShape(n_vocab=2048, n_embd=512, n_mult=256, n_head=8, n_layer=2, n_ctx=1200)
    m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
    om = orc.OracleModel(oshape); om.fill_synthetic()
    rng = np.random.default_rng(1100)
    toks = np.concatenate([[1], rng.integers(3, 2048, 1102)]).astype(np.int32)
    lp = m.prefill(toks[:1100], 0)
    for i in range(1100):
        lo, _ = om.eval(int(toks[i]), i)
    assert np.abs(lp - lo).max() < LOGIT_TOL
    for i in (1100, 1101):
        lg, _ = m.eval([int(toks[i])], i); lo, _ = om.eval(int(toks[i]), i)
        assert np.abs(lg - lo).max() < LOGIT_TOL
        assert int(lg.argmax()) == orc.greedy(lo)
    m.close()


def test_long_f32_cache_takes_16_wave_attention_workgroups(thk, orc, ctx):
    """attn_waves = 0 (auto, round 6): an f32 cache longer than 1024 rows with D = 128 runs attention with 16 waves per workgroup (a 256-position split = one round,
    every wave's single K/V batch in flight at once).  One 7B-wide layer, a 1536-row cache, 1100 prompt tokens through the slab prefill, then decode steps at
    T = 1101.. against the ORACLE fed token by token - and the same steps with attn_waves = 8 (the pipelined two-round kernel) agree to rounding."""
    shape = thk.ModelShape(n_vocab=2048, n_embd=4096, n_mult=256, n_head=32, n_layer=1, n_ctx=1536)
    oshape = orc.ModelShape(n_vocab=2048, n_embd=4096, n_mult=256, n_head=32, n_layer=1, n_ctx=1536)
    rng = np.random.default_rng(1536)
    toks = np.concatenate([[1], rng.integers(3, 2048, 1103)]).astype(np.int32)
    out = {}
    for aw in (0, 8):
        ctx.set_tunable("attn_waves", aw)
        try:
            m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
        finally:
            ctx.set_tunable("attn_waves", 0)
        m.prefill(toks[:1100], 0, want_logits=False)
        out[aw] = [m.eval([int(toks[i])], i)[0] for i in (1100, 1101, 1102)]
        m.close()
    om = orc.OracleModel(oshape); om.fill_synthetic()
    with fast_oracle(orc, True) as fl:
        for i in range(1100):
            om.eval(int(toks[i]), i, want_logits=False, flags=fl)
        for k, i in enumerate((1100, 1101, 1102)):
            lo, _ = om.eval(int(toks[i]), i, flags=fl)
            assert np.abs(out[0][k] - lo).max() < LOGIT_TOL, (i, float(np.abs(out[0][k] - lo).max()))
            assert int(out[0][k].argmax()) == orc.greedy(lo)
            assert np.abs(out[0][k] - out[8][k]).max() < 2e-5
    om.close()


@pytest.mark.parametrize("dims", [(512, 8, 2), (4096, 32, 2)], ids=["tiny-width", "7B-width"])
def test_prefill_deferred_norm_equals_explicit_norm(thk, orc, ctx, dims):
    """The 9-launch layer (RMSNorm's 1/rms applied on the output side of the GEMM, the residual reducers write the next image and
    the fixed-point sum of squares) against the 11-launch layer with its norm -> image launches (prefill_deferred_norm = 0) and
    against the oracle fed token by token: small-magnitude residual streams (embedding scale 0.02) are exactly where an
    un-prescaled image would lose the lo half of the hi/lo split to f16 subnormals."""
    E, H, L = dims
    shape = thk.ModelShape(n_vocab=2048, n_embd=E, n_mult=256, n_head=H, n_layer=L, n_ctx=256)
    oshape = orc.ModelShape(n_vocab=2048, n_embd=E, n_mult=256, n_head=H, n_layer=L, n_ctx=256)
    m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
    om = orc.OracleModel(oshape); om.fill_synthetic()
    rng = np.random.default_rng(E)
    toks = np.concatenate([[1], rng.integers(3, 2048, 150)]).astype(np.int32)
    for i in range(150):
        lo, _ = om.eval(int(toks[i]), i, flags=0)
    out = {}
    old = ctx.get_tunable("prefill_deferred_norm")
    try:
        for d in (1, 0):
            ctx.set_tunable("prefill_deferred_norm", d)      # read per prefill call
            m.reset_kv(0)
            out[d] = m.prefill(toks[:150], 0)                 # two slabs: 128 + 22 tokens
    finally:
        ctx.set_tunable("prefill_deferred_norm", old)
    assert np.abs(out[1] - lo).max() < 1e-4 and np.abs(out[0] - lo).max() < 1e-4
    assert np.abs(out[1] - out[0]).max() < 5e-5 and int(out[1].argmax()) == int(out[0].argmax()) == orc.greedy(lo)
    m.close(); om.close()


@pytest.mark.parametrize("M", [20, 128, 300])
def test_prefill_is_bitwise_repeatable(thk, ctx, M):
    """Race screen for the hand-synchronised LDS-DMA pipeline (counted vmcnt + raw barriers) and the stream-K
    reducers: the same prompt 60 times must give bit-identical logits (a DMA that is read before it lands shows up
    as a sporadic difference, not as a crash)."""
    shape = thk.ModelShape(n_layer=2)
    m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
    rng = np.random.default_rng(M)
    toks = np.concatenate([[1], rng.integers(3, 32000, M - 1)]).astype(np.int32)
    first = m.prefill(toks, 0).copy()
    for _ in range(60):
        m.reset_kv(0)
        again = m.prefill(toks, 0)
        assert np.array_equal(first.view(np.uint32), again.view(np.uint32))
    m.close()


@pytest.mark.parametrize("dims", [(512, 256, 8), (4096, 256, 32)])
@pytest.mark.parametrize("M", [7, 76, 128, 204])
def test_prefill_tile_images_equal_row_major(thk, dims, M):
    """The prefill GEMM streams repacked tile images of the layer matrices (tunable prefill_packed, the default) or the
    row-major matrices themselves: same MFMA order, so the logits must agree bit for bit.  The narrow model gives every
    workgroup a one-chunk share (all but one of its pipeline steps issue dummy loads) — the shape that exposed a dummy
    load landing in a live LDS stage during development."""
    E, mult, H = dims
    shape = thk.ModelShape(n_vocab=2048, n_embd=E, n_mult=mult, n_head=H, n_layer=2, n_ctx=256)
    rng = np.random.default_rng(M + E)
    toks = np.concatenate([[1], rng.integers(3, 2048, M - 1)]).astype(np.int32)
    out = []
    for packed in (0, 1):
        with thk.Context(0) as c:
            c.set_tunable("prefill_packed", packed)
            m = thk.Model(c, shape); m.fill_synthetic(); m.finalize()
            first = m.prefill(toks, 0).copy()
            for _ in range(5):
                m.reset_kv(0)
                assert np.array_equal(first.view(np.uint32), m.prefill(toks, 0).view(np.uint32))
            m.reset_kv(0)
            ld, _ = m.eval(toks, 0)
            assert np.abs(first - ld).max() < 2e-4          # vs the decode path, token by token
            out.append(first)
            m.close()
    assert np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32))


def test_prefill_tile_images_follow_weight_updates(thk, ctx):
    """The tile images are a second copy of the layer weights made on the first prefill call: writing a tensor afterwards
    (set_tensor or a new fill_synthetic) must invalidate them, or prefill would keep multiplying by the old matrix."""
    shape = thk.ModelShape(n_vocab=2048, n_embd=512, n_mult=256, n_head=8, n_layer=2, n_ctx=256)
    toks = np.concatenate([[1], np.random.default_rng(5).integers(3, 2048, 127)]).astype(np.int32)
    m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
    assert not m.prefill_uses_tile_images()
    m.prepare_prefill()                                             # explicit: the images (and the workspace) are paid for here ...
    assert m.prefill_uses_tile_images()
    before = m.prefill(toks, 0).copy()                              # ... not inside the first prefill call
    ctx.set_tunable("prefill_packed", 0)
    try:
        m.reset_kv(0)
        assert np.array_equal(m.prefill(toks, 0).view(np.uint32), before.view(np.uint32))     # row-major path: same bits
    finally:
        ctx.set_tunable("prefill_packed", 1)
    w = (np.random.default_rng(6).standard_normal((512, 512)) * 0.02).astype(np.float16)
    m.set_tensor("layers.1.attention.wo.weight", w.view(np.uint16))
    m.reset_kv(0)
    after = m.prefill(toks, 0).copy()
    m.reset_kv(0)
    ld, _ = m.eval(toks, 0)                                         # decode path reads the row-major matrix
    assert np.abs(after - ld).max() < 2e-4
    assert np.abs(after - before).max() > 1e-3                      # the update is visible at all
    m.fill_synthetic(); m.reset_kv(0)                               # back to the synthetic weights
    assert np.array_equal(m.prefill(toks, 0).view(np.uint32), before.view(np.uint32))
    m.close()


@pytest.mark.parametrize("dims", [(512, 8), (4096, 32), (5120, 40)], ids=["tiny-width", "7B-width", "13B-width"])
@pytest.mark.parametrize("M,n_past", [(129, 0), (200, 5), (256, 0), (300, 0)])
def test_prefill_256_token_slabs_equal_128_token_slabs(thk, dims, M, n_past):
    """Round 5: a slab is up to 256 tokens (eight token tiles; the GEMM multiplies every weight chunk against two token halves: gemm_prefill_v3h_kernel,
    accumulators = all 256 AccVGPRs), so a long prompt makes half the weight passes; `prefill_slab_tokens = 128` is the rounds-1-4 form.  Same arithmetic per
    token (each token is its own MFMA column; the K cut of the stream-K shares is the same): logits within 5e-5 of the 128-token slabs, within 2e-4 of the
    decode path, bit-identical over repeats; 129 / 200 tokens = one eight-tile pass with pad tiles, 300 = 256 + 44, n_past > 0 = attention over earlier rows
    (the slab's two attention launches of 128 queries write one image)."""
    E, H = dims
    shape = thk.ModelShape(n_vocab=2048, n_embd=E, n_mult=256, n_head=H, n_layer=2, n_ctx=320)
    rng = np.random.default_rng(M + E)
    toks = np.concatenate([[1], rng.integers(3, 2048, n_past + M)]).astype(np.int32)
    out, nxt = {}, {}
    for sl in (128, 256):
        with thk.Context(0) as c:
            c.set_tunable("prefill_slab_tokens", sl)
            m = thk.Model(c, shape); m.fill_synthetic(); m.finalize()
            if n_past:
                m.eval(toks[:n_past], 0, want_logits=False)
            first = m.prefill(toks[n_past:n_past + M], n_past).copy()
            for _ in range(3):
                m.reset_kv(0)
                if n_past:
                    m.eval(toks[:n_past], 0, want_logits=False)
                assert np.array_equal(first.view(np.uint32), m.prefill(toks[n_past:n_past + M], n_past).view(np.uint32))
            nxt[sl], _ = m.eval([int(toks[n_past + M])], n_past + M)
            if sl == 256:
                m.reset_kv(0)
                ld, _ = m.eval(toks[:n_past + M], 0)
                assert np.abs(first - ld).max() < 2e-4            # vs the decode path, token by token
            out[sl] = first
            m.close()
    assert np.abs(out[256] - out[128]).max() < 5e-5 and int(out[256].argmax()) == int(out[128].argmax())
    assert np.abs(nxt[256] - nxt[128]).max() < 5e-5


@pytest.mark.parametrize("dims", [(512, 8), (4096, 32), (5120, 40)], ids=["tiny-width", "7B-width", "13B-width"])
def test_prefill_wave_grid_equals_round5_slab_kernel_bit_for_bit(thk, dims):
    """gemm_prefill_v3g_kernel (round 6: a wave owns half the tile's rows x half the slab's tokens, half-steps by k-step) accumulates every output element in the order
    gemm_prefill_v3h_kernel does (ks0.hi ks0.lo ks1.hi ks1.lo per chunk, chunks in share order, shares summed in workgroup order by the same reducers): the logits of a
    300-token prompt (one full slab + a 44-token tail) and of a 200-token one (pad tiles) are the same BITS with prefill_wave_grid = 1 and 0."""
    E, H = dims
    shape = thk.ModelShape(n_vocab=2048, n_embd=E, n_mult=256, n_head=H, n_layer=2, n_ctx=320)
    rng = np.random.default_rng(E)
    with thk.Context(0) as ctx:
        m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
        for M in (300, 200):
            toks = np.concatenate([[1], rng.integers(3, 2048, M - 1)]).astype(np.int32)
            out = {}
            for wg in (1, 0):
                ctx.set_tunable("prefill_wave_grid", wg)
                m.reset_kv(0)
                out[wg] = m.prefill(toks, 0)
            ctx.set_tunable("prefill_wave_grid", 1)
            assert np.array_equal(out[0], out[1]), (dims, M, float(np.abs(out[0] - out[1]).max()))
        m.close()


def test_prefill_256_token_slab_with_f16_kv_cache_and_generic_attention(thk):
    """The eight-tile slab with the binary16 K/V cache option (the reducer rounds the rows it appends, attention widens them) and with the generic causal
    attention (`prefill_attn_mfma = 0`: attn_body with 200 queries, then an image launch of eight tiles): both against 128-token slabs."""
    shape = thk.ModelShape(n_vocab=2048, n_embd=512, n_mult=256, n_head=8, n_layer=2, n_ctx=320)
    toks = np.concatenate([[1], np.random.default_rng(77).integers(3, 2048, 210)]).astype(np.int32)
    for tun in ({"kv_f16": 1}, {"prefill_attn_mfma": 0}, {"kv_f16": 1, "prefill_attn_mfma": 0}):
        out = {}
        for sl in (128, 256):
            with thk.Context(0) as c:
                for k, v in tun.items():
                    c.set_tunable(k, v)
                c.set_tunable("prefill_slab_tokens", sl)
                m = thk.Model(c, shape); m.fill_synthetic(); m.finalize()
                m.eval(toks[:7], 0, want_logits=False)
                lp = m.prefill(toks[7:207], 7).copy()
                ln, _ = m.eval([int(toks[207])], 207)
                out[sl] = (lp, ln)
                m.close()
        assert np.abs(out[256][0] - out[128][0]).max() < 5e-5 and np.abs(out[256][1] - out[128][1]).max() < 5e-5, tun


def test_prefill_into_second_sequence_and_faithful_head(thk, orc, ctx):
    """Prefill writes the KV rows of the sequence it is given (not sequence 0) and honours the lm-head mode."""
    m, om = make_pair(thk, orc, ctx, "TINY_Q1", n_seq=2, lm_mode=1)     # 1 = THK_LMHEAD_FAITHFUL (defect Q1 reproduced)
    rng = np.random.default_rng(77)
    toks = [1] + rng.integers(3, 32000, 24).tolist()
    m.eval([5], 0, seq=0)                                               # sequence 0 holds unrelated state
    lp = m.prefill(toks, 0, seq=1)
    for i, t in enumerate(toks):
        lo, _ = om.eval(t, i, flags=orc.FAITHFUL_ORDER | orc.LM_FAITHFUL)
    assert np.abs(lp - lo).max() < LOGIT_TOL
    lg, _ = m.eval([9], len(toks), seq=1); lo2, _ = om.eval(9, len(toks), flags=orc.FAITHFUL_ORDER | orc.LM_FAITHFUL)
    assert np.abs(lg - lo2).max() < LOGIT_TOL
    m.close()


def test_prefill_13b_geometry(thk, ctx):
    """Prefill at the 13B row geometry (E=5120, H=40, F=13824; 2 layers): different row-block counts, K-chunk
    counts and stream-K shares than 7B.  Reference: token-by-token decode of the same prompt."""
    shape = thk.ModelShape(n_embd=5120, n_head=40, n_layer=2)
    a = thk.Model(ctx, shape); a.fill_synthetic(); a.finalize()
    b = thk.Model(ctx, shape); b.fill_synthetic(); b.finalize()
    rng = np.random.default_rng(13)
    toks = np.concatenate([[1], rng.integers(3, 32000, 99)]).astype(np.int32)      # 100 tokens: MT = 4 with 28 pad rows
    lp = a.prefill(toks, 0)
    ld, _ = b.eval(toks, 0)
    assert np.abs(lp - ld).max() < LOGIT_TOL
    assert int(lp.argmax()) == int(ld.argmax())
    a.close(); b.close()


def test_prefill_full_width_128_tokens(thk, orc, ctx):
    """128-token prompt at 7B row geometry (2 layers): MFMA GEMMs at (M=128, C=4096/11008) vs token-by-token decode."""
    shape = thk.ModelShape(n_layer=2)
    a = thk.Model(ctx, shape); a.fill_synthetic(); a.finalize()
    b = thk.Model(ctx, shape); b.fill_synthetic(); b.finalize()
    rng = np.random.default_rng(128)
    toks = np.concatenate([[1], rng.integers(3, 32000, 127)]).astype(np.int32)
    lp = a.prefill(toks, 0)
    ld, _ = b.eval(toks, 0)
    assert np.abs(lp - ld).max() < LOGIT_TOL
    assert int(lp.argmax()) == int(ld.argmax())
    la, _ = a.eval([77], 128); lb, _ = b.eval([77], 128)
    assert np.abs(la - lb).max() < LOGIT_TOL
    a.close(); b.close()


# ------------------------------------------------------------------ position contract / graphs / hooks
def test_decode_steps_refuse_to_run_past_n_ctx(thk, orc, ctx):
    """an advancing decode loop may not walk off the cache.  The last slot can be evaluated (and advanced from)
    once; anything further is THK_ERR_INVALID with nothing enqueued, until the sequence is re-positioned."""
    shape = thk.ModelShape(n_vocab=2048, n_embd=512, n_mult=256, n_head=8, n_layer=2, n_ctx=16)
    m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
    m.seq_set(0, 1, 0)
    m.decode_steps(16, 0, advance=True)                       # positions 0..15: exactly fills the context
    gen, n, pos = m.seq_get(0)
    assert n == 16 and pos == 15                              # the device never advances past n_ctx - 1
    with pytest.raises(thk.ThkError, match="n_ctx"):
        m.decode_step(0, advance=True)
    with pytest.raises(thk.ThkError, match="n_ctx"):
        m.decode_steps(3, 0, advance=False)
    _, n2, _ = m.seq_get(0)
    assert n2 == 16                                           # nothing ran
    m.seq_set(0, 7, 10)
    with pytest.raises(thk.ThkError, match="n_ctx"):
        m.decode_steps(7, 0, advance=True)                    # 10..16 would touch position 16
    m.decode_steps(6, 0, advance=True)                        # 10..15 is fine
    _, n3, pos3 = m.seq_get(0)
    assert n3 == 6 and pos3 == 15
    m.reset_kv(0)
    m.decode_step(0, advance=True)                            # usable again after a reset
    m.close()


def test_profile_step_holds_the_position(thk, ctx):
    """thk_model_profile_step is a hold-position step whatever the sequence's advance setting was, restores that setting, and
    is refused like any other step when the position has no cache row left - the host's position mirror stays exact."""
    shape = thk.ModelShape(n_vocab=2048, n_embd=512, n_mult=256, n_head=8, n_layer=2, n_ctx=16)
    m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
    m.seq_set(0, 1, 0)
    m.decode_steps(5, 0, advance=True)                        # leaves the advance flag set
    _, n0, p0 = m.seq_get(0)
    prof = m.profile_step(0)
    assert [k for k, _ in prof][:2] == ["norm_qkv_rope_kv", "attn_decode"] and all(ms >= 0 for _, ms in prof)
    _, _, p1 = m.seq_get(0)
    assert p1 == p0 == 5
    m.decode_step(0, advance=True)                            # the advance setting came back
    _, _, p2 = m.seq_get(0)
    assert p2 == 6
    m.decode_steps(10, 0, advance=True)                       # 6..15: the context is full, the mirror says so
    with pytest.raises(thk.ThkError, match="n_ctx"):
        m.decode_step(0, advance=True)
    m.close()


def test_prepare_steps_only_captures(thk, orc, ctx):
    """thk_model_prepare_steps builds the 8/4/2-step graphs without running anything; results equal single steps."""
    a, om = make_pair(thk, orc, ctx, "TINY")
    b = thk.Model(ctx, thk.TINY); b.fill_synthetic(); b.finalize()
    a.seq_set(0, 1, 0); b.seq_set(0, 1, 0)
    a.prepare_steps(15)
    _, n0, p0 = a.seq_get(0)
    assert n0 == 0 and p0 == 0
    a.decode_steps(15, 0, advance=True)                       # 8 + 4 + 2 + 1
    for _ in range(15):
        b.decode_step(0, advance=True)
    ga, na, pa = a.seq_get(0); gb, nb, pb = b.seq_get(0)
    assert na == nb == 15 and pa == pb == 15 and ga.tolist() == gb.tolist()
    a.close(); b.close(); om.close()


def test_measure_hook_is_refused_without_env(thk, ctx, monkeypatch):
    """the shipped ABI cannot be told to skip work unless the process opted in through the environment."""
    monkeypatch.delenv("THK_MEASURE_HOOKS", raising=False)
    with pytest.raises(thk.ThkError, match="THK_MEASURE_HOOKS"):
        ctx.set_tunable("measure_skip_kernel", 4)
    assert ctx.get_tunable("measure_skip_kernel") == 0
    ctx.set_tunable("measure_skip_kernel", 0)                 # clearing is always allowed
    monkeypatch.setenv("THK_MEASURE_HOOKS", "1")
    ctx.set_tunable("measure_skip_kernel", 4)
    ctx.set_tunable("measure_skip_kernel", 0)


# ------------------------------------------------------------------ optional f16 KV cache
KV16_TOL = 1e-3   # vs the oracle that rounds k, v to binary16 at the append: same tolerance as the f32 path


@pytest.mark.parametrize("use_graph", [0, 1])
def test_f16_kv_cache_vs_oracle_rounded_at_append(thk, orc, ctx, use_graph):
    """tunable kv_f16 = 1: caches stored as binary16 (RNE at the append), decode attention widens them on load.  Checked
    against the oracle with K/V rounded to f16 at the append (flag KV_F16); the default stays f32 like the reference
    (th-llama-loader.cpp:335) and the two really differ."""
    m, om = make_pair(thk, orc, ctx, "TINY", tunables={"kv_f16": 1, "use_graph": use_graph})
    ref = thk.Model(ctx, thk.TINY); ref.fill_synthetic(); ref.finalize()
    rng = np.random.default_rng(16)
    toks = [1] + rng.integers(3, 2048, 40).tolist()
    differs = 0.0
    for i, t in enumerate(toks):
        lg, _ = m.eval([t], i)
        lo, _ = om.eval(t, i, flags=orc.FAITHFUL_ORDER | orc.KV_F16)
        l32, _ = ref.eval([t], i)
        assert np.abs(lg - lo).max() < KV16_TOL, i
        differs = max(differs, float(np.abs(lg - l32).max()))
    assert differs > 1e-5                       # the option is really on (f16 rounding of K/V is visible in the logits)
    T = 41
    assert m.bytes_per_token(T) == thk.TINY.bytes_per_token(T, kv_bytes=2) < ref.bytes_per_token(T) == thk.TINY.bytes_per_token(T, kv_bytes=4)
    m.close(); ref.close(); om.close()


def test_f16_kv_prefill_and_decode_continue(thk, orc, ctx):
    """Prefill writes the f16 cache (RNE, same as the decode append) and its MFMA attention reads it; decode continues on it."""
    m, om = make_pair(thk, orc, ctx, "TINY", tunables={"kv_f16": 1})
    rng = np.random.default_rng(17)
    toks = [1] + rng.integers(3, 2048, 40).tolist()
    lp = m.prefill(toks[:33], 0)
    for i in range(33):
        lo, _ = om.eval(toks[i], i, flags=orc.KV_F16)
    assert np.abs(lp - lo).max() < KV16_TOL and int(lp.argmax()) == orc.greedy(lo)
    for i in range(33, 41):
        lg, _ = m.eval([toks[i]], i); lo, _ = om.eval(toks[i], i, flags=orc.KV_F16)
        assert np.abs(lg - lo).max() < KV16_TOL, i
    m.close(); om.close()


def test_f16_kv_full_width_7b_layers(thk, orc, ctx):
    """7B row geometry (D = 128, H = 32) with the f16 cache at a few hundred positions of context."""
    shape = thk.ModelShape(n_embd=4096, n_head=32, n_layer=1)
    ctx.set_tunable("kv_f16", 1)
    try:
        m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
    finally:
        ctx.set_tunable("kv_f16", 0)
    om = orc.OracleModel(orc.ModelShape(n_embd=4096, n_head=32, n_layer=1)); om.fill_synthetic()
    rng = np.random.default_rng(18)
    toks = [1] + rng.integers(3, 32000, 7).tolist()
    for i, t in enumerate(toks):
        lg, _ = m.eval([t], i); lo, _ = om.eval(t, i, flags=orc.KV_F16)
        assert np.abs(lg - lo).max() < KV16_TOL, i
    assert not m.uses_engine()
    m.close(); om.close()


# ------------------------------------------------------------------ full-size configurations (BASELINE.json configs 3 and 5)
def test_7b_full_model_prefill_128_vs_token_by_token(thk, ctx):
    """BASELINE config 3 at FULL size: the 128-token prompt through all 32 layers of the MFMA prefill path gives the
    logits of the same prompt fed token by token through the decode path (prefill parity is defined against decode,
    SURVEY.md Q5), the same greedy token, and a cache from which decode continues to the same next logits."""
    shape = thk.LLAMA_7B
    rng = np.random.default_rng(128)
    toks = np.concatenate([[1], rng.integers(3, shape.n_vocab, 128)]).astype(np.int32)
    a = thk.Model(ctx, shape); a.fill_synthetic(); a.finalize()
    lp = a.prefill(toks[:128], 0)
    la_next, _ = a.eval([int(toks[128])], 128)
    a.close()
    b = thk.Model(ctx, shape); b.fill_synthetic(); b.finalize()
    ld, _ = b.eval(toks[:128], 0)
    lb_next, _ = b.eval([int(toks[128])], 128)
    b.close()
    assert np.isfinite(lp).all()
    assert np.abs(lp - ld).max() < LOGIT_TOL and int(lp.argmax()) == int(ld.argmax())
    assert np.abs(la_next - lb_next).max() < LOGIT_TOL


def test_7b_round4_step_equals_round3_step(thk, ctx):
    """Full 7B, real geometry: the round-4 step (attention splits over the live context, greedy pick folded into the lm-head
    launch, kernel arguments preloaded) against round 3's launch geometry (splits over n_ctx, finish_token launch).  The default model free-runs 24 greedy tokens on the
    device; the round-3 geometry is teacher-forced with them: logits equal to summation-order noise at every position, the same
    pick wherever the top-2 margin is not itself noise; the same at the last cache slot."""
    shape = thk.LLAMA_7B
    rng = np.random.default_rng(4)
    prompt = np.concatenate([[1], rng.integers(3, shape.n_vocab, 40)]).astype(np.int32)
    P = len(prompt)

    def build(tun):
        old = {k: ctx.get_tunable(k) for k in tun}
        for k, v in tun.items():
            ctx.set_tunable(k, v)
        try:
            m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
        finally:
            for k, v in old.items():
                ctx.set_tunable(k, v)
        return m
    a = build({})
    la, _ = a.eval(prompt, 0)
    first = int(la.argmax())
    a.seq_set(0, first, P)
    a.decode_steps(24, 0, advance=True)
    gen = a.seq_get(0)[0].tolist()
    la_end = a.read_logits(0)                         # logits of the 24th free-running step
    a.seq_set(0, 5, 511)
    a.decode_steps(1, 0, advance=False)
    la_last, tok_last = a.read_logits(0), a.seq_get(0)[0].tolist()
    a.close()
    b = build(R03_STEP)
    lb, _ = b.eval(prompt, 0)
    assert np.abs(la - lb).max() < 1e-4 and int(lb.argmax()) == first
    toks = [first] + gen
    worst = 0.0
    for i in range(24):
        lb, _ = b.eval([toks[i]], P + i)
        top2 = np.sort(lb)[-2:]
        if top2[1] - top2[0] > 1e-3:
            assert int(lb.argmax()) == gen[i], (i, int(lb.argmax()), gen[i])
    worst = float(np.abs(lb - la_end).max())
    assert worst < 2e-4, worst
    b.seq_set(0, 5, 511)
    b.decode_steps(1, 0, advance=False)
    lb_last = b.read_logits(0)
    assert np.abs(la_last - lb_last).max() < 2e-4
    top2 = np.sort(lb_last)[-2:]
    assert top2[1] - top2[0] < 1e-3 or b.seq_get(0)[0].tolist() == tok_last
    b.close()


def test_13b_full_model_properties(thk, ctx):
    """BASELINE config 5 at FULL size (LLaMA-13B, 25.7 GB of synthetic f16 weights, E=5120 H=40 L=40 F=13824) at T=512:
    determinism, hold-position idempotence at the last cache slot, device loop == eval-path greedy tokens, two chained
    half-model stages == the full model (mirrors test_7b_full_model_properties)."""
    shape = thk.LLAMA_13B
    m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
    rng = np.random.default_rng(13)
    prompt = [1] + rng.integers(3, 32000, 11).tolist()
    lg1, h1 = m.eval(prompt, 0, want_hidden=True)
    assert np.isfinite(lg1).all() and np.isfinite(h1).all()
    m.reset_kv(0)
    lg2, _ = m.eval(prompt, 0)
    assert (lg1 == lg2).all()
    toks_host, lg = [], lg1
    for i in range(3):
        t = int(lg.argmax()); toks_host.append(t)
        lg, _ = m.eval([t], len(prompt) + i)
    m.reset_kv(0)
    m.eval(prompt[:-1], 0, want_logits=False)
    m.seq_set(0, prompt[-1], len(prompt) - 1)
    m.decode_steps(3, 0, advance=True)
    gen, n, pos = m.seq_get(0)
    assert gen.tolist() == toks_host and pos == len(prompt) + 2
    m.seq_set(0, 5, 511)
    m.decode_step(0, advance=False); a, _, _ = m.seq_get(0)
    m.seq_set(0, 5, 511)
    m.decode_step(0, advance=False); b, _, _ = m.seq_get(0)
    assert a[0] == b[0]
    assert m.bytes_per_token(512) == 26545377280          # SURVEY.md 8(d) canonical 13B figure
    m.close()
    s0 = thk.Model(ctx, shape, 0, 20, flags=thk.THK_STAGE_EMBED); s0.fill_synthetic(); s0.finalize()
    s1 = thk.Model(ctx, shape, 20, 40, flags=thk.THK_STAGE_HEAD); s1.fill_synthetic(); s1.finalize()
    for i, t in enumerate(prompt):
        _, h = s0.eval([t], i, want_logits=False, want_hidden=True)
        lgp, _ = s1.eval(None, i, hidden=h)
    assert np.abs(lgp - lg1).max() < 1e-5
    s0.close(); s1.close()
