"""GPU parity at FULL DEPTH: every layer of LLaMA-7B (32) / LLaMA-13B (40) through the HIP decode path against the oracle's
fast flavour, on the seeded synthetic model and the very prompt bench.py times (seed 511).  The north_star's claim is "logits
within 1e-3 on the same prompt" for the whole model - error growth over the full depth and over T = 512 cache rows is measured
here, not extrapolated from 1-2 layer models (replaces th_eval_gpu, th-llama.cpp:464-660, end to end).

Slow by GPU-suite standards (the oracle decodes ~20-45 tokens/s on the box's host cores: 519 oracle tokens for 7B) but bounded:
one oracle model at a time, skipped with a reason when host RAM cannot hold it."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3   # north_star: logits within 1e-3
PREFILL_TOKENS = 128   # BASELINE config C3
# prompt lengths pushed through thk_model_prefill and compared with the ORACLE at their last position: 128 = config C3 (one 128-token slab,
# gemm_prefill_v3_kernel), 256 = one full 256-token slab (gemm_prefill_v3h_kernel + the slab attention), 511 = 256 + 255 (the pad-tile
# path of the second slab, attention over cached rows of the first) = the prompt behind bench.py's prompt_512_tokens_ms
PREFILL_LENGTHS = (PREFILL_TOKENS, 256, 511)


def _mem_available():
    try:
        return [int(l.split()[1]) * 1024 for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0]
    except Exception:
        return 0


def _oracle_bytes(oshape):
    E, F, V = oshape.n_embd, oshape.n_ff, oshape.n_vocab
    return oshape.n_layer * (4 * E * E + 3 * E * F) * 2 + 2 * V * E * 2 + oshape.n_layer * 2 * oshape.n_ctx * E * 4


def _bench_prompt(n_vocab, T, seq=0):
    """bench.py synthetic_prompt(): BOS + ids uniform in [3, n_vocab), seed 511 + sequence index."""
    rng = np.random.default_rng(511 + seq)
    return np.concatenate([[1], rng.integers(3, n_vocab, T - 1)]).astype(np.int32)


def _full_depth(thk, orc, ctx, name, early, T):
    shape, oshape = getattr(thk, name), getattr(orc, name)
    need = int(_oracle_bytes(oshape) * 1.25)
    if _mem_available() < need:
        pytest.skip(f"host RAM: {_mem_available() / 2**30:.0f} GiB available < {need / 2**30:.0f} GiB for the oracle's {name} model")
    threads = orc.num_threads()
    orc.set_num_threads(orc.usable_cpus())
    m = om = mf = None
    try:
        m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
        om = orc.OracleModel(oshape); om.fill_synthetic()
        prompt = _bench_prompt(shape.n_vocab, T)
        worst_l = worst_h = 0.0
        t0 = time.time()
        # positions 0 .. early-1: logits, final hidden state and the greedy pick at every one
        for i in range(early):
            lg, hid = m.eval([int(prompt[i])], i, want_hidden=True)
            lo, ho = om.eval(int(prompt[i]), i, flags=0)
            dl, dh = float(np.abs(lg - lo).max()), float(np.abs(hid - ho).max())
            worst_l, worst_h = max(worst_l, dl), max(worst_h, dh)
            assert dl < LOGIT_TOL, (name, i, dl)
            assert dh < LOGIT_TOL * max(1.0, float(np.abs(ho).max())), (name, i, dh)
            assert int(lg.argmax()) == orc.greedy(lo), (name, i)
        # both lm-head modes at one position (Q1): the faithful combine on a second HIP model
        mf = thk.Model(ctx, shape); mf.fill_synthetic(); mf.set_lmhead_mode(thk.THK_LMHEAD_FAITHFUL); mf.finalize()
        lgf = None
        for i in range(2):
            lgf, _ = mf.eval([int(prompt[i])], i)
        lof, _ = om.eval(int(prompt[1]), 1, flags=orc.LM_FAITHFUL)     # rewrites cache row 1 with the same values
        assert np.abs(lgf - lof).max() < LOGIT_TOL
        skipped = orc.q1_skipped_indices(shape.n_vocab)
        lo1, _ = om.eval(int(prompt[1]), 1, flags=0)
        assert np.abs(lgf[skipped] - lo1[skipped]).max() > 1e-3          # the defect is visible at full depth, and only there:
        keep = np.ones(shape.n_vocab, bool); keep[skipped] = False
        assert np.abs(lgf[keep] - lo1[keep]).max() < LOGIT_TOL
        mf.close(); mf = None
        # the rest of the prompt: n_past = early .. T-2 (the cache fill bench.py does), then the timed position n_past = T-1
        m.eval(prompt[early:T - 1], early, want_logits=False)
        want = {}
        for i in range(early, T - 1):
            if i + 1 in PREFILL_LENGTHS:
                want[i + 1] = om.eval(int(prompt[i]), i, flags=0)        # what a prefill of i + 1 tokens must reproduce
            else:
                om.eval(int(prompt[i]), i, want_logits=False, flags=0)
        # config C3 at full depth against the ORACLE (not against the HIP decode path): the first 128 tokens of the very prompt through
        # the MFMA prefill path (stream-K GEMMs + reducers, MFMA attention) on a second HIP model - logits and the final
        # hidden state at position 127 (semantics th-llama.cpp:464-660 with the batch branch :307-311)
        # round 6: the 256-token slab path (the default for every prompt > 128 tokens) against the oracle at the same widths and depth
        mp = thk.Model(ctx, shape); mp.fill_synthetic(); mp.finalize()
        try:
            for n in PREFILL_LENGTHS:
                if n > T - 1:
                    continue
                mp.reset_kv(0)
                lp = mp.prefill(prompt[:n], 0)
                hp = mp.debug_buffer("x")
                lo_pf, ho_pf = want[n]
                dpl, dph = float(np.abs(lp - lo_pf).max()), float(np.abs(hp - ho_pf).max())
                print(f"\n[full-depth {name}] {n}-token prefill vs oracle at position {n - 1}: max |dlogit| {dpl:.3e} (hidden {dph:.3e})")
                assert dpl < LOGIT_TOL, (name, "prefill", n, dpl)
                assert dph < LOGIT_TOL * max(1.0, float(np.abs(ho_pf).max())), (name, "prefill", n, dph)
                assert int(lp.argmax()) == orc.greedy(lo_pf), (name, "prefill", n)
        finally:
            mp.close()
        lg, hid = m.eval([int(prompt[T - 1])], T - 1, want_hidden=True)
        lo, ho = om.eval(int(prompt[T - 1]), T - 1, flags=0)
        dl, dh = float(np.abs(lg - lo).max()), float(np.abs(hid - ho).max())
        print(f"\n[full-depth {name}] max |dlogit| positions 0..{early - 1}: {worst_l:.3e} (hidden {worst_h:.3e}); at n_past={T - 1}: {dl:.3e} "
              f"(hidden {dh:.3e}, |logit| max {np.abs(lo).max():.2f}); oracle on {orc.num_threads()} threads, {time.time() - t0:.1f}s")
        assert dl < LOGIT_TOL, (name, T - 1, dl)
        assert dh < LOGIT_TOL * max(1.0, float(np.abs(ho).max())), (name, T - 1, dh)
        assert int(lg.argmax()) == orc.greedy(lo)
        # the device-resident greedy loop (what bench.py times: hold position at n_past = T-1) picks the oracle's token
        # (a hold-position step feeds its pick back as the next input token at the same slot, so only the first step sees prompt[T-1])
        m.seq_set(0, int(prompt[T - 1]), T - 1)
        m.decode_steps(3, 0, advance=False)
        gen, n, pos = m.seq_get(0)
        assert n == 3 and pos == T - 1 and int(gen[0]) == orc.greedy(lo)
        lo2, _ = om.eval(int(gen[0]), T - 1, flags=0)
        assert int(gen[1]) == orc.greedy(lo2)
    finally:
        orc.set_num_threads(threads)
        for x in (m, mf, om):
            if x is not None:
                x.close()


def test_7b_full_depth_logits_vs_oracle(thk, orc, ctx):
    """32-layer LLaMA-7B: positions 0..7 and n_past = 511 after bench.py's 511-token prompt."""
    _full_depth(thk, orc, ctx, "LLAMA_7B", early=8, T=512)


def test_13b_full_depth_logits_vs_oracle(thk, orc, ctx):
    """40-layer LLaMA-13B (config C5): positions 0..7 and n_past = 511 after the same seeded prompt."""
    _full_depth(thk, orc, ctx, "LLAMA_13B", early=8, T=512)
