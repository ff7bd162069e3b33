"""GPU tests of the C++ host layer: the kept GGML loader + TokenHawk host API over libthk.
A synthetic ggjt v1 file (tiny model, toy vocabulary) is written, loaded through
load_llama_file / the streamed capi_* path, and checked against the CPU oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import ggjt

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(thk):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "token-hawk_amd", "host")])
    lib = C.CDLL(os.path.join(ROOT, "token-hawk_amd", "libthk_host.so"))
    lib.thh_last_error.restype = C.c_char_p
    lib.capi_last_error.restype = C.c_char_p
    lib.capi_transcript.restype = C.c_char_p
    lib.capi_on_human_message.restype = None
    lib.capi_on_human_message.argtypes = [C.c_char_p]
    lib.capi_model_begin_load.restype = None
    lib.capi_load_model_header.restype = None
    lib.capi_load_model_weights.restype = None
    lib.capi_model_end_load.restype = C.c_bool
    lib.capi_set_context.argtypes = [C.c_void_p]
    lib.thh_set_greedy_device_loop.argtypes = [C.c_int64, C.c_int]
    lib.thh_tensor_buffer_semantics.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    lib.thh_load_file.restype = C.c_int64
    lib.thh_load_file.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.thh_eval.argtypes = [C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.thh_do_inference.argtypes = [C.c_int64, C.c_char_p, C.c_void_p, C.c_char_p, C.c_int]
    lib.thh_set_sampler.argtypes = [C.c_int64, C.c_int, C.c_float, C.c_float, C.c_float]
    lib.thh_set_prefill.argtypes = [C.c_int64, C.c_int]
    lib.thh_set_device_topk.argtypes = [C.c_int64, C.c_int]
    lib.thh_free.argtypes = [C.c_int64]; lib.thh_reset.argtypes = [C.c_int64]; lib.thh_hparams.argtypes = [C.c_int64, C.c_void_p]
    lib.capi_model_begin_load.argtypes = []
    lib.capi_load_model_header.argtypes = [C.c_char_p, C.c_double]
    lib.capi_load_model_weights.argtypes = [C.c_char_p, C.c_double, C.c_double]
    lib.capi_set_sampler.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float]
    lib.thh_make_synthetic.restype = C.c_int64
    lib.thh_make_synthetic.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_float]
    lib.thh_set_step_limit.argtypes = [C.c_int64, C.c_int64]
    lib.thh_collect_stats.argtypes = [C.c_int64, C.c_int]
    lib.thh_stats.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
    lib.thh_device_model.restype = C.c_void_p; lib.thh_device_model.argtypes = [C.c_int64]
    lib.capi_device_model.restype = C.c_void_p; lib.capi_device_model.argtypes = []
    return lib


@pytest.fixture(scope="module")
def model_file(orc, tmp_path_factory):
    path = str(tmp_path_factory.mktemp("ggjt") / "tiny-f16.bin")
    offs, words, scores = ggjt.write_synthetic_model(path, orc, orc.TINY)
    return path, offs, words, scores


def host_shape(orc):
    """The ggjt file carries no n_ctx: the host model uses LlamaModel's default 512 (th-llama.hpp:105)."""
    s = orc.TINY
    return orc.ModelShape(s.n_vocab, s.n_embd, s.n_mult, s.n_head, s.n_layer, 512)


def greedy_reference(orc, words, prompt_ids, n_ctx=512, max_steps=500, start_pos=0, om=None):
    """do_inference semantics: <= 500 evaluations per message (prompt tokens included), stop on EOS or n_ctx."""
    if om is None:
        om = orc.OracleModel(host_shape(orc)); om.fill_synthetic()
    pos, out, steps = start_pos, [], 0
    for t in prompt_ids:
        lg, _ = om.eval(t, pos); pos += 1; steps += 1
    tok = orc.greedy(lg)
    while tok != 2:
        out.append(tok)
        if steps >= max_steps or pos >= n_ctx:
            break
        lg, _ = om.eval(tok, pos); pos += 1; steps += 1
        tok = orc.greedy(lg)
    return out, pos


def test_load_file_and_eval_match_oracle(host, orc, ctx, model_file):
    path, _, words, scores = model_file
    h = host.thh_load_file(ctx.h, path.encode(), 0)
    assert h > 0, host.thh_last_error()
    hp = (C.c_int32 * 7)(); host.thh_hparams(h, hp)
    s = orc.TINY
    assert list(hp)[:5] == [s.n_vocab, s.n_embd, s.n_mult, s.n_head, s.n_layer]
    host.thh_set_sampler(h, 40, 0.95, 0.0, 1.1)        # greedy branch of the sampler
    om = orc.OracleModel(host_shape(orc)); om.fill_synthetic()
    logits = np.empty(s.n_vocab, np.float32)
    for i, t in enumerate([1, 300, 17, 5]):
        ids = np.array([t], np.int32)
        tok = host.thh_eval(h, C.c_void_p(ids.ctypes.data), 1, i, C.c_void_p(logits.ctypes.data))
        lo, _ = om.eval(t, i)
        assert np.abs(logits - lo).max() < 1e-3
        assert tok == orc.greedy(lo)
    host.thh_free(h)


def test_do_inference_greedy_matches_oracle(host, orc, ctx, model_file):
    path, _, words, scores = model_file
    h = host.thh_load_file(ctx.h, path.encode(), 0)
    assert h > 0, host.thh_last_error()
    host.thh_set_sampler(h, 40, 0.95, 0.0, 1.1)
    n_past = C.c_int32(); text = C.create_string_buffer(1 << 16)
    n_new = host.thh_do_inference(h, b"hello world", C.byref(n_past), text, len(text))
    # reference semantics: ' ' is prepended on a fresh context, BOS added (th-llama.cpp:121-125)
    import test_host_cpu as thc
    ids = thc.py_tokenize(words, scores, b" hello world", True)
    exp, pos = greedy_reference(orc, words, ids)
    assert n_new == len(exp)
    assert text.value == b"".join(words[t] for t in exp)
    assert n_past.value == pos
    # the context is now at the reference's 500-position guard: a further message is refused (th-llama.cpp:112-119)
    n2 = host.thh_do_inference(h, b"x", C.byref(n_past), text, len(text))
    assert n2 == 0 and n_past.value == pos and b"Maximum context reached" in host.thh_last_error()
    host.thh_reset(h)
    n3 = host.thh_do_inference(h, b"hello world", C.byref(n_past), text, len(text))
    assert n3 == len(exp) and n_past.value == pos       # reset really clears the KV state
    host.thh_free(h)


UPDATE_FN = C.CFUNCTYPE(None, C.c_char_p, C.c_char_p)
SEND_FN = C.CFUNCTYPE(None, C.c_char_p, C.c_char_p)


def test_streamed_capi_load_matches_file_load(host, orc, ctx, model_file):
    """The reference's wasm exports with their own signatures (web/main.cpp:83-179): capi_model_begin_load() /
    capi_load_model_header(data, size) / capi_load_model_weights(data, offset, size) / bool capi_model_end_load() /
    void capi_on_human_message(str).  The file is fed header first, then one tensor record at a time; the reply streams
    through the two UI hooks while capi_on_human_message has already returned (non-blocking, as in the browser)."""
    path, offs, words, scores = model_file
    blob = open(path, "rb").read()
    hp = (C.c_int32 * 7)(); consumed = C.c_int64(); nv = C.c_int32()
    assert host.thh_parse_header(blob, C.c_int64(len(blob)), hp, C.byref(consumed), C.byref(nv)) == 1
    updates, bots = [], []
    upd = UPDATE_FN(lambda mid, text: updates.append((mid, text)))
    snd = SEND_FN(lambda text, mid: bots.append((text, mid)))
    host.capi_set_context(ctx.h)
    host.capi_set_ui_hooks(upd, snd)
    host.capi_model_begin_load()
    host.capi_load_model_header(blob[:consumed.value], float(consumed.value))
    pos = consumed.value
    name = C.create_string_buffer(128); ty = C.c_int32(); shape = (C.c_int64 * 4)(); ne = (C.c_int64 * 2)()
    a, b, rb = C.c_int64(), C.c_int64(), C.c_int64()
    while pos < len(blob):
        rec = blob[pos:]
        assert host.thh_parse_tensor(rec, C.c_int64(len(rec)), C.c_int64(pos), name, 128, C.byref(ty), shape, ne, C.byref(a), C.byref(b), C.byref(rb)) == 1
        host.capi_load_model_weights(blob[pos:pos + rb.value], float(pos), float(rb.value))
        pos += rb.value
    assert host.capi_model_end_load() is True, host.capi_last_error()
    host.capi_set_sampler(40, 0.95, 0.0, 1.1)
    host.capi_on_human_message(b"hello world")
    host.capi_on_human_message(b"ignored: a reply is still being generated")     # web/main.cpp:172
    host.capi_wait_idle()
    assert host.capi_inference_complete() == 1
    out = host.capi_transcript()
    import test_host_cpu as thc
    exp, _ = greedy_reference(orc, words, thc.py_tokenize(words, scores, b" hello world", True))
    assert out == b"".join(words[t] for t in exp)
    assert bots[0] == (b"--", b"bot-msg-1") and len([x for x in bots if x[0] == b"--"]) == 1       # the second message opened no reply
    assert len(updates) == len(exp) and updates[-1] == (b"bot-msg-1", out) and all(m == b"bot-msg-1" for m, _ in updates)
    host.capi_on_human_message(b"[cmd] reset")
    assert bots[-1][0] == b"LLM context reset."
    host.capi_on_human_message(b"hello world"); host.capi_wait_idle()
    assert host.capi_transcript() == out
    host.capi_model_unload()
    host.capi_set_ui_hooks(None, None)


def test_capi_load_failure_is_reported_by_end_load(host, ctx, model_file):
    """A bad header makes capi_model_end_load return false (the void loaders cannot report it themselves)."""
    host.capi_set_context(ctx.h)
    host.capi_model_begin_load()
    host.capi_load_model_header(b"not a ggjt file at all", 22.0)
    assert host.capi_model_end_load() is False
    assert b"magic" in host.capi_last_error()
    host.capi_model_unload()


def test_greedy_device_loop_matches_eval_path(host, ctx, model_file):
    """with temp <= 0 do_inference generates in the device-resident loop (4-byte token read-backs per 8
    steps) instead of a 128 KB logits read-back per token: same text, same token count, same final position, with and
    without prompt prefill."""
    path, _, words, scores = model_file
    out = []
    for loop, prefill in ((0, 0), (1, 0), (1, 1)):
        h = host.thh_load_file(ctx.h, path.encode(), 0)
        assert h > 0, host.thh_last_error()
        host.thh_set_sampler(h, 40, 0.95, 0.0, 1.1)
        host.thh_set_greedy_device_loop(h, loop); host.thh_set_prefill(h, prefill)
        n_past = C.c_int32(); text = C.create_string_buffer(1 << 16)
        n_new = host.thh_do_inference(h, b"the quick brown fox", C.byref(n_past), text, len(text))
        n2 = host.thh_do_inference(h, b"and then", C.byref(n_past), text, len(text))          # a follow-up message continues the context
        out.append((n_new, n2, n_past.value, text.value))
        host.thh_free(h)
    assert out[0][0] > 0 and out[0] == out[1] == out[2]


def test_tensor_buffer_semantics_on_device(host, ctx):
    """A18: TensorBuffer (th.hpp:83-148) is what load_weights uploads every tensor through; here its allocation, upload,
    move construction / assignment (source emptied, same device allocation), shape relabelling and download are checked."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((7, 256)).astype(np.float32)
    back = np.zeros_like(x)
    ok = host.thh_tensor_buffer_semantics(ctx.h, x.ctypes.data, 7, 256, back.ctypes.data)
    assert ok == 0x3F, bin(ok)
    assert (back == x).all()


def test_missing_tensor_is_reported(host, orc, ctx, tmp_path):
    words, scores = ggjt.toy_vocab(orc.TINY.n_vocab)
    s = orc.TINY
    hp = dict(n_vocab=s.n_vocab, n_embd=s.n_embd, n_mult=s.n_mult, n_head=s.n_head, n_layer=s.n_layer, n_rot=64, ftype=1)
    tensors = [t for t in ggjt.synthetic_model_tensors(orc, s) if t[0] != "layers.1.feed_forward.w2.weight"]
    path = str(tmp_path / "broken.bin")
    ggjt.write_ggjt(path, hp, words, scores, tensors)
    assert host.thh_load_file(ctx.h, path.encode(), 0) == 0
    assert host.thh_load_file(ctx.h, b"/nonexistent/model.bin", 0) == 0


def test_cli_greedy(host, orc, model_file):
    path, _, words, scores = model_file
    exe = os.path.join(ROOT, "token-hawk_amd", "thk_cli")
    r = subprocess.run([exe, "-m", path, "--greedy", "hello world"], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr
    import test_host_cpu as thc
    exp, _ = greedy_reference(orc, words, thc.py_tokenize(words, scores, b" hello world", True))
    assert r.stdout == b"".join(words[t] for t in exp)
    assert subprocess.run([exe, "-d", "dir", "x"], capture_output=True).returncode == 2


@pytest.mark.parametrize("temp", [0.0, 0.8])
def test_prompt_prefill_generates_the_same_text(host, ctx, model_file, temp):
    """do_inference with prefillPrompt (one thk_model_prefill call for the whole prompt) == the reference-style loop
    that feeds the prompt one token per step: same text, same final position, for the greedy branch and for the
    seeded top-k/top-p sampler (the discarded per-prompt-token draws are replayed on the random stream)."""
    path, _, words, scores = model_file
    out = []
    for prefill in (0, 1):
        h = host.thh_load_file(ctx.h, path.encode(), 0)       # fresh model => fresh mt19937(780658349)
        assert h > 0, host.thh_last_error()
        host.thh_set_sampler(h, 40, 0.95, temp, 1.1)
        host.thh_set_prefill(h, prefill)
        n_past = C.c_int32(); text = C.create_string_buffer(1 << 16)
        n_new = host.thh_do_inference(h, b"the quick brown fox jumps over the lazy dog", C.byref(n_past), text, len(text))
        out.append((n_new, n_past.value, text.value))
        host.thh_free(h)
    assert out[0][0] > 0
    assert out[0] == out[1]


@pytest.mark.parametrize("params", [(40, 0.95, 0.8, 1.1), (5, 1.0, 1.5, 1.3), (100, 0.6, 0.4, 1.0)])
def test_device_topk_sampler_generates_the_same_text(host, ctx, model_file, params):
    """Stochastic do_inference with the sampler's candidates selected on the device (thk_model_logits_topk: k x 8 bytes per token)
    == the reference-style path that reads all n_vocab logits back per token (th-llama.cpp:686-724): same seed, same text, same
    final position, across two messages (the second one runs with a filled repetition-penalty window)."""
    path, _, words, scores = model_file
    k, p, t, pen = params
    out = []
    for dev_topk in (0, 1):
        h = host.thh_load_file(ctx.h, path.encode(), 0)       # fresh model => fresh mt19937(780658349)
        assert h > 0, host.thh_last_error()
        host.thh_set_sampler(h, k, p, t, pen)
        host.thh_set_device_topk(h, dev_topk)
        n_past = C.c_int32(); text = C.create_string_buffer(1 << 16)
        n1 = host.thh_do_inference(h, b"the quick brown fox", C.byref(n_past), text, len(text))
        t1 = text.value
        n2 = host.thh_do_inference(h, b"jumps over", C.byref(n_past), text, len(text))
        out.append((n1, n2, n_past.value, t1, text.value))
        host.thh_free(h)
    assert out[0][0] > 0
    assert out[0] == out[1]


@pytest.mark.parametrize("k", [1, 41, 64, 300, 1024])
def test_model_eval_topk_equals_eval_plus_sort(thk, ctx, k):
    """thk_model_eval_topk (the stochastic sampler's per-token call: step + two-launch top-k + keys written into host-mapped memory + polled stamp) against
    thk_model_eval's full logits sorted on the host (value descending, ties by ascending id), over several positions and multi-token calls; k = 1024 at
    V = 32000 takes the top-k's two merge levels."""
    m = thk.Model(ctx, thk.TINY_Q1); m.fill_synthetic(); m.finalize()
    r = thk.Model(ctx, thk.TINY_Q1); r.fill_synthetic(); r.finalize()
    toks = [1, 5, 9, 31999, 77]
    pos = 0
    for chunk in ([toks[0]], toks[1:3], [toks[3]], [toks[4]]):
        vals, ids = m.eval_topk(chunk, pos, k)
        for i, t in enumerate(chunk):       # the reference side one step at a time: eval_topk steps its tokens singly too
            lg, _ = r.eval([t], pos + i)
        order = np.lexsort((np.arange(lg.size), -lg))[:k]
        assert np.array_equal(ids, order.astype(np.int32)) and np.array_equal(vals.view(np.uint32), lg[order].view(np.uint32)), (pos, k)
        pos += len(chunk)
    m.close(); r.close()


def test_model_logits_topk_and_read_logits(thk, ctx):
    m = thk.Model(ctx, thk.TINY_Q1); m.fill_synthetic(); m.finalize()
    lg, _ = m.eval([1, 5, 9], 0)
    full = m.read_logits()
    assert (full.view(np.uint32) == lg.view(np.uint32)).all()
    vals, ids = m.logits_topk(64)
    order = np.lexsort((np.arange(lg.size), -lg.astype(np.float64)))[:64]
    assert ids.tolist() == order.tolist() and (vals == lg[order]).all()
    m.close()


def test_cli_matches_library(host, ctx, model_file):
    """`thk_cli -m file --greedy "prompt"` (the reference's `th -m ... "prompt"` contract, cli/main.cpp:182-198) prints
    the text do_inference produces through the library, with and without --prefill."""
    import subprocess
    path, _, words, scores = model_file
    cli = os.path.join(ROOT, "token-hawk_amd", "thk_cli")
    if not os.path.exists(cli):
        pytest.skip("thk_cli not built")
    h = host.thh_load_file(ctx.h, path.encode(), 0)
    host.thh_set_sampler(h, 40, 0.95, 0.0, 1.1)
    n_past = C.c_int32(); text = C.create_string_buffer(1 << 16)
    host.thh_do_inference(h, b"hello world", C.byref(n_past), text, len(text))
    host.thh_free(h)
    for extra in ([], ["--prefill"]):
        r = subprocess.run([cli, "-m", path, "--greedy", *extra, "hello world"], capture_output=True, timeout=240)
        assert r.returncode == 0, r.stderr.decode()[-400:]
        assert r.stdout == text.value
        assert b"tokens generated" in r.stderr
    r = subprocess.run([cli, "--bogus"], capture_output=True, timeout=60)
    assert r.returncode == 2 and b"unknown option" in r.stderr


@pytest.mark.parametrize("temp", [0.8, 0.0])
def test_timed_generation_equals_untimed_and_honours_the_step_limit(host, thk, orc, ctx, model_file, temp):
    """bench.py's extras.host_api: th::do_inference limited to prompt + n steps (LlamaModel::stepLimit) with the per-section timers on
    (LlamaModel::collectStats) generates exactly the text of the untimed call, stops after prompt + n steps, and accounts for every
    step; the same generation on a LlamaModel made WITHOUT a file (thh_make_synthetic: device-side synthetic fill + the caller's
    vocabulary, what the bench uses for the 7B) gives the same text as the file-loaded model."""
    path, _, words, scores = model_file
    prompt = b"the quick brown fox"
    import test_host_cpu as thc
    n_prompt = len(thc.py_tokenize(words, scores, b" " + prompt, True))
    n_new = 24
    got = []
    for how, timed in (("file", 0), ("file", 1), ("synthetic", 1)):
        if how == "file":
            h = host.thh_load_file(ctx.h, path.encode(), 0)
        else:
            s = orc.TINY
            hp6 = np.array([s.n_vocab, s.n_embd, s.n_mult, s.n_head, s.n_layer, 512], np.int32)
            blob = b"".join(words); lens = np.array([len(w) for w in words], np.int32)
            h = host.thh_make_synthetic(ctx.h, hp6.ctypes.data, blob, lens.ctypes.data, np.asarray(scores, np.float32).ctypes.data, thk.TENSOR_SEED, thk.TENSOR_SIGMA)
        assert h > 0, host.thh_last_error()
        host.thh_set_sampler(h, 40, 0.95, temp, 1.1)
        host.thh_set_step_limit(h, n_prompt + n_new)
        host.thh_collect_stats(h, timed)
        n_past = C.c_int32(); text = C.create_string_buffer(1 << 16)
        n_tok = host.thh_do_inference(h, prompt, C.byref(n_past), text, len(text))
        out8 = (C.c_double * 8)(); ends = (C.c_double * 1024)()
        n_steps = host.thh_stats(h, out8, ends, 1024)
        if timed:
            ends = np.array(ends[:n_steps])
            assert n_steps == n_past.value and (np.diff(ends) >= 0).all()
            if temp > 0:
                assert int(out8[4]) == n_steps and int(out8[5]) + int(out8[6]) >= n_steps      # one eval per step; top-k or the full read-back after each
        else:
            assert n_steps == 0
        got.append((n_tok, n_past.value, text.value))
        host.thh_free(h)
    assert got[0][1] <= n_prompt + n_new and got[0][0] >= 1
    assert got[0] == got[1] == got[2]


def _avail(path="/proc/meminfo"):
    try:
        return [int(l.split()[1]) * 1024 for l in open(path) if l.startswith("MemAvailable")][0]
    except Exception:
        return 0


def test_7b_sized_ggjt_file_through_the_kept_loader(host, thk, orc, ctx, tmp_path_factory):
    """Config C1 at FULL size (north_star keeps "the GGML-f16 loader"): the synthetic LLaMA-7B written as a real ggjt v1 file
    (291 tensors, 32000-entry vocabulary, 13.5 GB) goes through load_llama_file (th-llama-loader.cpp:485-635: record by record through
    a host buffer, TensorBuffer upload, device-to-device hand-over) and through the streamed capi_* path (web/main.cpp:83-104); the
    loaded weights are compared byte for byte with the device-side synthetic fill (three whole tensors + every norm vector), the
    logits of three positions and the greedy tokens bit for bit with the fill-synthetic model.  Reports the load rate."""
    import mmap
    import shutil
    import time
    oshape = orc.LLAMA_7B
    file_bytes = sum(int(np.prod(shp)) * (2 if dt == "f16" else 4) for _, dt, shp in oshape.tensor_specs())
    shm = "/dev/shm"
    use_shm = os.path.isdir(shm) and shutil.disk_usage(shm).free > file_bytes + (2 << 30)
    base = shm if use_shm else str(tmp_path_factory.mktemp("ggjt7b"))
    if not use_shm and shutil.disk_usage(base).free < file_bytes + (2 << 30):
        pytest.skip(f"no place for a {file_bytes / 2**30:.1f} GiB model file (/dev/shm and the temp directory are too small)")
    need = (file_bytes if use_shm else 0) + (6 << 30)            # the file's pages (tmpfs) + generator / loader buffers
    if _avail() < need:
        pytest.skip(f"host RAM: {_avail() / 2**30:.0f} GiB available < {need / 2**30:.0f} GiB for a 7B-sized ggjt file in {base}")
    path = os.path.join(base, f"thk-synth-7b-f16-{os.getpid()}.bin")
    threads = orc.num_threads()
    orc.set_num_threads(orc.usable_cpus())
    a = None
    try:
        t0 = time.time()
        offs, words, scores = ggjt.write_synthetic_model(path, orc, oshape)
        t_write = time.time() - t0
        size = os.path.getsize(path)
        assert len(offs) == 3 + 9 * oshape.n_layer == 291 and size > file_bytes
        a = thk.Model(ctx, thk.LLAMA_7B); a.fill_synthetic(); a.finalize()
        toks = [1, 17, 400]
        want = [a.eval([t], i)[0].copy() for i, t in enumerate(toks)]
        E, F, V = oshape.n_embd, oshape.n_ff, oshape.n_vocab
        whole = [("tok_embeddings.weight", np.uint16, V * E), ("layers.0.attention.wq.weight", np.uint16, E * E),
                 ("layers.31.feed_forward.w2.weight", np.uint16, E * F), ("output.weight", np.uint16, V * E), ("norm.weight", np.float32, E)]
        whole += [(f"layers.{l}.{n}_norm.weight", np.float32, E) for l in range(oshape.n_layer) for n in ("attention", "ffn")]
        lib = ctx.lib

        def check(dev, how):
            for name, dt, n in whole:
                got = np.empty(n, dt)
                ctx.check(lib.thk_model_get_tensor(dev, name.encode(), 0, got.nbytes, got.ctypes.data), "thk_model_get_tensor")
                assert np.array_equal(got, a.get_tensor(name, dt, n)), (how, name)
            for name in ("layers.7.feed_forward.w1.weight", "layers.19.attention.wo.weight", "layers.31.feed_forward.w3.weight"):   # head and tail of some more
                n = E * F if "feed_forward" in name else E * E
                for off in (0, n - 65536):
                    got = np.empty(65536, np.uint16)
                    ctx.check(lib.thk_model_get_tensor(dev, name.encode(), off * 2, got.nbytes, got.ctypes.data), "thk_model_get_tensor")
                    assert np.array_equal(got, a.get_tensor(name, np.uint16, 65536, off)), (how, name, off)

        # ---- load_llama_file
        t0 = time.time()
        h = host.thh_load_file(ctx.h, path.encode(), 0)
        t_load = time.time() - t0
        assert h > 0, host.thh_last_error()
        hp = (C.c_int32 * 7)(); host.thh_hparams(h, hp)
        assert list(hp)[:5] == [V, E, oshape.n_mult, oshape.n_head, oshape.n_layer]
        check(host.thh_device_model(h), "load_llama_file")
        host.thh_set_sampler(h, 40, 0.95, 0.0, 1.1)                 # greedy: th_eval reads the logits back
        lg = np.empty(V, np.float32)
        for i, t in enumerate(toks):
            tok = host.thh_eval(h, (C.c_int32 * 1)(t), 1, i, lg.ctypes.data)
            assert np.array_equal(lg.view(np.uint32), want[i].view(np.uint32)) and tok == int(want[i].argmax()), ("load_llama_file", i)
        host.thh_free(h)
        # ---- the streamed capi_* path: header, then one tensor record at a time straight from the mapped file
        t0 = time.time()
        with open(path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
            view = np.frombuffer(mm, np.uint8)
            base_addr = view.ctypes.data
            hp7 = (C.c_int32 * 7)(); consumed = C.c_int64(); nv = C.c_int32()
            assert host.thh_parse_header(C.c_void_p(base_addr), C.c_int64(min(size, 8 << 20)), hp7, C.byref(consumed), C.byref(nv)) == 1
            host.capi_set_context(ctx.h)
            host.capi_model_begin_load()
            host.capi_load_model_header(C.cast(base_addr, C.c_char_p), float(consumed.value))
            pos, n_rec = consumed.value, 0
            name = C.create_string_buffer(128); ty = C.c_int32(); shape4 = (C.c_int64 * 4)(); ne = (C.c_int64 * 2)()
            d0, d1, rb = C.c_int64(), C.c_int64(), C.c_int64()
            while pos < size:
                assert host.thh_parse_tensor(C.c_void_p(base_addr + pos), C.c_int64(size - pos), C.c_int64(pos), name, 128, C.byref(ty), shape4, ne,
                                             C.byref(d0), C.byref(d1), C.byref(rb)) == 1
                host.capi_load_model_weights(C.cast(base_addr + pos, C.c_char_p), float(pos), float(rb.value))
                pos += rb.value; n_rec += 1
            assert n_rec == 291
            assert host.capi_model_end_load() is True, host.capi_last_error()
            del view
        t_stream = time.time() - t0
        dev = host.capi_device_model()
        check(dev, "capi_*")
        for i, t in enumerate(toks):
            ctx.check(lib.thk_model_eval(dev, 0, (C.c_int32 * 1)(t), 1, i, None, lg.ctypes.data), "thk_model_eval")
            assert np.array_equal(lg.view(np.uint32), want[i].view(np.uint32)), ("capi_*", i)
        host.capi_model_unload()
        print(f"\n[C1 full size] {size / 1e9:.2f} GB ggjt v1 file in {base}: written in {t_write:.1f} s (oracle generator, {orc.num_threads()} threads); "
              f"load_llama_file {t_load:.1f} s = {size / t_load / 1e9:.2f} GB/s; streamed capi_* load {t_stream:.1f} s = {size / t_stream / 1e9:.2f} GB/s; "
              f"{len(whole)} tensors byte-identical to the device-side fill, logits of {len(toks)} positions bit-identical")
    finally:
        orc.set_num_threads(threads)
        if a is not None:
            a.close()
        try:
            os.remove(path)
        except OSError:
            pass
