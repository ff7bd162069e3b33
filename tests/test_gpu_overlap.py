"""Overlapped dispatch (tunable overlap_dispatch, token-hawk_amd/csrc/thk_ovl.cpp): the decode step's launches as AQL packets on
a queue of libthk's own, chosen packets without the barrier bit and the dependency enforced inside the kernels.  Same kernels'
bodies, same arithmetic: the hipGraph path is the reference for bit-level equality, the oracle for parity (th-llama.cpp:464-660)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3      # same tolerance as the launch path against the oracle (tests/test_gpu_model.py)
KEEP_MASKS = [118, 0, 127, 96, 111]      # default | everything overlapped | every packet a barrier packet | only head + pick ordered by barriers | only w2 waits inside


def _greedy(m, ctx, overlap, n, prompt_token=1):
    ctx.set_tunable("overlap_dispatch", overlap)
    m.reset_kv(0); m.seq_set(0, prompt_token, 0)
    m.decode_steps(n, 0, advance=True)
    gen, ng, pos = m.seq_get(0)
    lg = m.read_logits(0)
    ctx.set_tunable("overlap_dispatch", 0)
    assert ng == n and pos == n
    return gen[:ng].copy(), lg.copy()


@pytest.mark.parametrize("keep", KEEP_MASKS)
@pytest.mark.parametrize("E,H,L,steps", [(4096, 32, 2, 24), (5120, 40, 1, 12)])
def test_overlap_equals_graph_path(thk, ctx, E, H, L, steps, keep):
    """Greedy tokens and final logits of an advancing decode loop: overlapped dispatch == hipGraph replays, for every choice of
    which packets keep the barrier bit (the four kernel flavours: plain, waiting, arriving, both)."""
    old = ctx.get_tunable("overlap_keep_barrier")
    ctx.set_tunable("overlap_keep_barrier", keep)
    try:
        m = thk.Model(ctx, thk.ModelShape(n_embd=E, n_head=H, n_layer=L)); m.fill_synthetic(); m.finalize()
        ref_t, ref_l = _greedy(m, ctx, 0, steps)
        ctx.set_tunable("overlap_dispatch", 1)
        assert m.uses_overlap()
        ctx.set_tunable("overlap_dispatch", 0)
        for _ in range(2):                                   # twice: the arrival counters must come back to zero after every step
            t, l = _greedy(m, ctx, 1, steps)
            assert (t == ref_t).all()
            assert np.abs(l - ref_l).max() < 1e-5
        m.close()
    finally:
        ctx.set_tunable("overlap_keep_barrier", old)


def test_overlap_vs_oracle_full_width(thk, orc, ctx):
    """7B row geometry, 2 layers, token by token through decode_step on the overlapped queue against the oracle."""
    shape = thk.ModelShape(n_embd=4096, n_head=32, n_layer=2)
    m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
    om = orc.OracleModel(orc.ModelShape(n_embd=4096, n_head=32, n_layer=2)); om.fill_synthetic()
    toks = [1] + np.random.default_rng(3).integers(3, 32000, 4).tolist()
    ctx.set_tunable("overlap_dispatch", 1)
    try:
        m.reset_kv(0)
        for i, t in enumerate(toks):
            m.seq_set(0, t, i)                               # a stream-ordered kernel between two batches of the private queue
            m.decode_step(0, advance=False)
            lg = m.read_logits(0)
            lo, _ = om.eval(t, i, flags=0)
            assert np.abs(lg - lo).max() < LOGIT_TOL, i
            assert int(lg.argmax()) == orc.greedy(lo)
            assert m.seq_last_token(0) == orc.greedy(lo)
    finally:
        ctx.set_tunable("overlap_dispatch", 0)
    m.close(); om.close()


def test_overlap_after_prefill_and_mixed_with_graph_steps(thk, ctx):
    """Stream-ordered meaning of the calls: a prefill (HIP kernels), then decode steps alternating between the two paths, equal a
    run that never leaves the hipGraph path."""
    shape = thk.ModelShape(n_embd=4096, n_head=32, n_layer=2)
    a = thk.Model(ctx, shape); a.fill_synthetic(); a.finalize()
    prompt = np.concatenate([[1], np.random.default_rng(9).integers(3, 32000, 37)]).astype(np.int32)
    def run(switch):
        a.reset_kv(0)
        lp = a.prefill(prompt, 0)
        a.seq_set(0, int(lp.argmax()), len(prompt))
        for k in range(6):
            ctx.set_tunable("overlap_dispatch", switch[k])
            a.decode_steps(3, 0, advance=True)
        ctx.set_tunable("overlap_dispatch", 0)
        gen, n, pos = a.seq_get(0)
        return gen[:n].copy(), a.read_logits(0)
    try:
        ref_t, ref_l = run([0] * 6)
        t, l = run([1, 0, 1, 1, 0, 1])
    finally:
        ctx.set_tunable("overlap_dispatch", 0)
    assert (t == ref_t).all() and np.abs(l - ref_l).max() < 1e-5
    a.close()


def test_overlap_f16_kv_and_two_sequences(thk, ctx):
    """The f16 cache flavour of the attention kernels, and two sequences sharing the queue and the counters."""
    ctx.set_tunable("kv_f16", 1)
    try:
        m = thk.Model(ctx, thk.ModelShape(n_embd=4096, n_head=32, n_layer=1), n_seq=2); m.fill_synthetic(); m.finalize()
    finally:
        ctx.set_tunable("kv_f16", 0)
    out = {}
    for mode in (0, 1):
        ctx.set_tunable("overlap_dispatch", mode)
        try:
            for s, tok in ((0, 1), (1, 77)):
                m.reset_kv(s); m.seq_set(s, tok, 0)
            for _ in range(5):
                m.decode_steps(2, 0, advance=True)
                m.decode_steps(2, 1, advance=True)
            out[mode] = [m.seq_get(s)[0][:10].copy() for s in (0, 1)] + [m.read_logits(s) for s in (0, 1)]
        finally:
            ctx.set_tunable("overlap_dispatch", 0)
    assert (out[0][0] == out[1][0]).all() and (out[0][1] == out[1][1]).all()
    assert np.abs(out[0][2] - out[1][2]).max() < 1e-5 and np.abs(out[0][3] - out[1][3]).max() < 1e-5
    m.close()


def test_overlap_refuses_models_without_overlapped_kernels(thk, ctx):
    """The tunable never falls back silently: a model the overlapped kernels do not cover makes the decode calls fail."""
    m = thk.Model(ctx, thk.TINY); m.fill_synthetic(); m.finalize()
    m.seq_set(0, 1, 0)
    ctx.set_tunable("overlap_dispatch", 1)
    try:
        assert not m.uses_overlap()
        with pytest.raises(thk.ThkError, match="overlap"):
            m.decode_step(0, advance=True)
        with pytest.raises(thk.ThkError, match="overlap"):
            m.decode_steps(4, 0, advance=True)
    finally:
        ctx.set_tunable("overlap_dispatch", 0)
    m.decode_step(0, advance=True)                          # and the stream-ordered path is untouched
    assert m.seq_get(0)[1] == 1
    m.close()


def test_7b_overlap_hold_position_same_token(thk, ctx):
    """Whole LLaMA-7B at n_past = 511, the workload bench.py times: both paths pick the same token, step after step."""
    m = thk.Model(ctx, thk.LLAMA_7B); m.fill_synthetic(); m.finalize()
    res = {}
    for mode in (0, 1):
        ctx.set_tunable("overlap_dispatch", mode)
        try:
            m.seq_set(0, 5, 511)
            m.decode_steps(24, 0, advance=False)
            res[mode] = (m.seq_last_token(0), m.read_logits(0))
        finally:
            ctx.set_tunable("overlap_dispatch", 0)
    assert res[0][0] == res[1][0]
    assert np.abs(res[0][1] - res[1][1]).max() < 1e-5
    m.close()


def test_debug_buffer_matches_between_paths(thk, ctx):
    """thk_model_debug_buffer (development aid): after ONE step from the same state the last layer's q, split partials, SwiGLU
    vector and the final hidden state are the same on both launch paths, and unknown names are refused."""
    m = thk.Model(ctx, thk.ModelShape(n_embd=4096, n_head=32, n_layer=1)); m.fill_synthetic(); m.finalize()
    bufs = {}
    for mode in (0, 1):
        ctx.set_tunable("overlap_dispatch", mode)
        try:
            m.reset_kv(0); m.seq_set(0, 1, 0)
            m.decode_step(0, advance=True)
            bufs[mode] = {n: m.debug_buffer(n) for n in ("q", "part_ml", "part_o", "u", "x")}
        finally:
            ctx.set_tunable("overlap_dispatch", 0)
    assert bufs[0]["q"].size == 4096 and bufs[0]["u"].size == 11008 and bufs[0]["part_o"].size == 32 * 4 * 128
    for n in bufs[0]:
        a, b = bufs[0][n], bufs[1][n]
        fin = np.isfinite(a)                                  # empty splits carry m = -inf
        assert (np.isfinite(b) == fin).all() and np.abs(a[fin] - b[fin]).max() < 1e-5, n
    with pytest.raises(thk.ThkError, match="no working buffer"):
        m.debug_buffer("nope")
    m.close()
