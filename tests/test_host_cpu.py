"""CPU tests of the C++ host layer (token-hawk_amd/host): fp16 converters, TensorShape,
tokenizer, sampler and ggjt parser, each against an independent Python restatement of the
reference algorithm (th.cpp:312-359, th-llama.cpp:802-1108, th-llama-loader.cpp:47-265).
No device is touched."""
import bisect
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

import ggjt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def P(a):
    """numpy array -> void* (ctypes would truncate a bare int address to 32 bits)."""
    return C.c_void_p(a.ctypes.data)


@pytest.fixture(scope="module")
def host(thk):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "token-hawk_amd", "host")])
    lib = C.CDLL(os.path.join(ROOT, "token-hawk_amd", "libthk_host.so"))
    lib.thh_last_error.restype = C.c_char_p
    lib.capi_test_capi.restype = C.c_char_p
    return lib


def test_capi_present(host):
    assert host.capi_test_capi() == b"thk host capi"
    for s in ("capi_model_begin_load", "capi_load_model_header", "capi_load_model_weights", "capi_model_end_load", "capi_on_human_message",
              "capi_set_context", "capi_set_ui_hooks", "capi_wait_idle", "capi_inference_complete", "capi_transcript"):
        assert hasattr(host, s)   # the reference's wasm exports, web/main.cpp:72-179


def test_fp16_converters_match_ieee(host):
    h = np.arange(65536, dtype=np.uint16)
    out = np.empty(65536, np.float32)
    host.thh_fp16_to_fp32(P(h), P(out), C.c_int64(h.size))
    ref = h.view(np.float16).astype(np.float32)
    fin = np.isfinite(ref)
    assert (out.view(np.uint32)[fin] == ref.view(np.uint32)[fin]).all()
    back = np.empty(65536, np.uint16)
    host.thh_fp32_to_fp16(P(out), P(back), C.c_int64(h.size))
    assert (back[fin] == h[fin]).all()
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s for s in (1e-8, 1e-4, 1.0, 3e4, 1e6)])
    got = np.empty(x.size, np.uint16)
    host.thh_fp32_to_fp16(P(x), P(got), C.c_int64(x.size))
    with np.errstate(over="ignore"):
        assert (got == x.astype(np.float16).view(np.uint16)).all()


@pytest.mark.parametrize("lbrc,canon,total", [((0, 0, 1, 4096), (0, 0, 1, 4096), 4096), ((1, 1, 0, 7), (0, 0, 1, 7), 7),
                                               ((0, 512, 32, 128), (0, 512, 32, 128), 512 * 32 * 128), ((0, 0, 0, 0), (0, 0, 1, 0), 0)])
def test_tensor_shape(host, lbrc, canon, total):
    out = (C.c_int64 * 5)()
    host.thh_tensor_shape_roundtrip(*[C.c_int64(v) for v in lbrc], out)
    assert tuple(out[:4]) == canon and out[4] == total


# ------------------------------------------------------------------ tokenizer
def py_tokenize(words, scores, text: bytes, add_bos):
    """Independent restatement: repeatedly merge the best-scoring adjacent pair (leftmost on ties)."""
    if not text:
        return []
    tok2id = {}
    for i, w in enumerate(words):
        tok2id[w] = i
    pieces, off = [], 0
    while off < len(text):
        n = [1] * 12 + [2, 2, 3, 4]
        ln = min(len(text) - off, n[text[off] >> 4])
        pieces.append(text[off:off + ln]); off += ln
    while True:
        best = None
        for i in range(len(pieces) - 1):
            cat = pieces[i] + pieces[i + 1]
            if cat in tok2id and (best is None or scores[tok2id[cat]] > best[0]):
                best = (scores[tok2id[cat]], i)
        if best is None:
            break
        i = best[1]
        pieces[i:i + 2] = [pieces[i] + pieces[i + 1]]
    out = [1] if add_bos else []
    for p in pieces:
        out += [tok2id[p]] if p in tok2id else [b + 3 for b in p]
    return out


def c_tokenize(host, words, scores, text: bytes, add_bos):
    blob = b"".join(words)
    lens = np.array([len(w) for w in words], np.int32)
    out = np.empty(len(text) + 2, np.int32)
    n = host.thh_tokenize(blob, P(lens), P(scores), len(words), text, len(text), int(add_bos), P(out), out.size)
    return out[:n].tolist()


@pytest.mark.parametrize("text", [" hello world", "hello", " hell", "héllo", "xyz\n", " ", "", "lololo", " hhheee"])
def test_tokenizer_matches_restatement(host, text):
    words, scores = ggjt.toy_vocab(400)
    for bos in (True, False):
        assert c_tokenize(host, words, scores, text.encode(), bos) == py_tokenize(words, scores, text.encode(), bos)


def test_tokenizer_properties(host):
    words, scores = ggjt.toy_vocab(400)
    ids = c_tokenize(host, words, scores, b" hello world", True)
    assert ids[0] == 1                                              # BOS
    assert ids[1:] == [words.index(b" hello"), words.index(b" world")]   # full merges win
    assert c_tokenize(host, words, scores, b"\xff", False) == [0xFF + 3]  # byte fallback id = byte + 3
    assert c_tokenize(host, words, scores, b"", True) == []           # empty text -> no BOS either (th-llama.cpp:1052-1054)
    rng = np.random.default_rng(1)
    for _ in range(50):
        s = bytes(rng.choice(list(b" helowrd"), size=int(rng.integers(1, 30))).tolist())
        assert c_tokenize(host, words, scores, s, False) == py_tokenize(words, scores, s, False)


# ------------------------------------------------------------------ sampler
class MT19937:
    def __init__(self, seed):
        self.mt = [0] * 624; self.idx = 624
        self.mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            self.mt[i] = (1812433253 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 30)) + i) & 0xFFFFFFFF

    def __call__(self):
        if self.idx >= 624:
            for i in range(624):
                y = (self.mt[i] & 0x80000000) | (self.mt[(i + 1) % 624] & 0x7FFFFFFF)
                self.mt[i] = self.mt[(i + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self.idx = 0
        y = self.mt[self.idx]; self.idx += 1
        y ^= y >> 11; y ^= (y << 7) & 0x9D2C5680; y ^= (y << 15) & 0xEFC60000; y ^= y >> 18
        return y & 0xFFFFFFFF


def py_sample(rng, logits, top_k, top_p, temp, penalty, last_n):
    """llama_sample_top_p_top_k restated (th-llama.cpp:814-907) incl. libstdc++'s discrete_distribution."""
    n = len(logits)
    if temp <= 0:
        return int(np.argmax(logits))
    scale = np.float32(1.0) / np.float32(temp)
    cand = []
    for i in range(n):
        v = np.float32(logits[i]) * scale
        if i in last_n:
            v = np.float32(logits[i]) * scale * np.float32(penalty) if logits[i] < 0 else np.float32(logits[i]) * scale / np.float32(penalty)
        cand.append((np.float32(v), i))
    if 0 < top_k < n:
        cand.sort(key=lambda t: -t[0]); cand = cand[:top_k]
    maxl = max(c[0] for c in cand)
    probs = [np.exp(np.float32(c[0] - maxl), dtype=np.float32) for c in cand]
    s = float(np.sum(np.array(probs, np.float64)))
    probs = [np.float32(p / s) for p in probs]
    if top_p < 1.0:
        cum = 0.0
        for i, p in enumerate(probs):
            cum += float(p)
            if cum >= top_p:
                probs, cand = probs[:i + 1], cand[:i + 1]; break
        inv = 1.0 / cum
        probs = [np.float32(float(p) * inv) for p in probs]
    if len(probs) < 2:
        return cand[0][1]
    pd = np.array(probs, np.float64); pd = pd / pd.sum()
    cp = np.cumsum(pd); cp[-1] = 1.0
    x1, x2 = rng(), rng()
    u = (float(x1) + float(x2) * 4294967296.0) / 18446744073709551616.0
    if u >= 1.0:
        u = np.nextafter(1.0, 0.0)
    return cand[bisect.bisect_left(cp.tolist(), u)][1]


@pytest.mark.parametrize("top_k,top_p,temp", [(40, 0.95, 0.8), (5, 1.0, 1.0), (0, 0.5, 1.3), (40, 0.95, 0.0)])
def test_sampler_matches_restatement(host, top_k, top_p, temp):
    rng = np.random.default_rng(top_k + int(temp * 10))
    logits = (rng.standard_normal(2000) * 2).astype(np.float32)
    seed, n_draws = 780658349, 25       # the reference's fixed seed (th-llama-loader.cpp:332-333)
    out = np.empty(n_draws, np.int32)
    last = np.array([3, 77, 1999], np.int32)
    host.thh_sample(C.c_uint32(seed), P(logits), logits.size, top_k, C.c_float(top_p), C.c_float(temp), C.c_float(1.1),
                    P(last), last.size, n_draws, P(out))
    mt = MT19937(seed)
    exp = [py_sample(mt, logits, top_k, top_p, temp, 1.1, set(last.tolist())) for _ in range(n_draws)]
    assert out.tolist() == exp


def test_sampler_greedy_first_max(host):
    lg = np.array([0.5, 2.0, 2.0, -1.0], np.float32)
    out = np.empty(1, np.int32)
    host.thh_sample(C.c_uint32(1), P(lg), 4, 40, C.c_float(0.95), C.c_float(0.0), C.c_float(1.1), None, 0, 1, P(out))
    assert out[0] == 1


# ------------------------------------------------------------------ ggjt parser
def test_header_and_tensor_records(host, orc, tmp_path):
    path = str(tmp_path / "tiny.bin")
    offs, words, scores = ggjt.write_synthetic_model(path, orc, orc.TINY)
    blob = open(path, "rb").read()
    hp = (C.c_int32 * 7)(); consumed = C.c_int64(); nv = C.c_int32()
    assert host.thh_parse_header(blob, C.c_int64(len(blob)), hp, C.byref(consumed), C.byref(nv)) == 1
    s = orc.TINY
    assert list(hp) == [s.n_vocab, s.n_embd, s.n_mult, s.n_head, s.n_layer, s.n_embd // s.n_head, 1] and nv.value == s.n_vocab
    pos, seen = consumed.value, []
    name = C.create_string_buffer(128); ty = C.c_int32(); shape = (C.c_int64 * 4)(); ne = (C.c_int64 * 2)()
    doff, dbytes, rbytes = C.c_int64(), C.c_int64(), C.c_int64()
    while pos < len(blob):
        rec = blob[pos:]
        assert host.thh_parse_tensor(rec, C.c_int64(len(rec)), C.c_int64(pos), name, 128, C.byref(ty), shape, ne, C.byref(doff),
                                     C.byref(dbytes), C.byref(rbytes)) == 1, host.thh_last_error()
        nm = name.value.decode()
        assert pos + doff.value == offs[nm] and offs[nm] % 32 == 0            # payload on a 32-byte file offset
        seen.append((nm, ty.value, tuple(shape), tuple(ne), dbytes.value))
        pos += rbytes.value
    assert pos == len(blob)
    specs = {n: (dt, shp) for n, dt, shp in s.tensor_specs()}
    assert [x[0] for x in seen] == [n for n, _, _ in s.tensor_specs()]
    for nm, ty_, shp, ne_, nbytes in seen:
        dt, want = specs[nm]
        assert ty_ == (1 if dt == "f16" else 0)
        if len(want) == 2:
            assert shp == (0, 0, want[0], want[1]) and ne_ == (want[1], want[0]) and nbytes == want[0] * want[1] * 2
        else:
            assert shp == (0, 0, 1, want[0]) and ne_ == (want[0], 1) and nbytes == want[0] * 4   # 1-D tensors are f32, canonicalised to r=1


def test_header_rejects_bad_files(host):
    hp = (C.c_int32 * 7)(); consumed = C.c_int64(); nv = C.c_int32()
    words, scores = ggjt.toy_vocab(10)

    def header(magic=ggjt.MAGIC_GGJT, version=1, n_vocab=10):
        b = struct.pack("<II7i", magic, version, n_vocab, 512, 256, 8, 2, 64, 1)
        for w, s in zip(words, scores):
            b += struct.pack("<I", len(w)) + w + struct.pack("<f", float(s))
        return b

    good = header()
    assert host.thh_parse_header(good, C.c_int64(len(good)), hp, C.byref(consumed), C.byref(nv)) == 1 and consumed.value == len(good)
    for bad, why in ((header(magic=ggjt.MAGIC_GGML), b"unversioned"), (header(magic=0x12345678), b"magic"), (header(version=2), b"version"),
                     (good[:40], b"runcated"), (header(n_vocab=0), b"hyper")):
        assert host.thh_parse_header(bad, C.c_int64(len(bad)), hp, C.byref(consumed), C.byref(nv)) == 0
        assert why in host.thh_last_error()
    # quantized tensor record is refused
    rec = struct.pack("<3i2i", 2, 1, 2, 32, 1) + b"q" + b"\0" * 64
    name = C.create_string_buffer(16); ty = C.c_int32(); shape = (C.c_int64 * 4)(); ne = (C.c_int64 * 2)(); a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
    assert host.thh_parse_tensor(rec, C.c_int64(len(rec)), C.c_int64(0), name, 16, C.byref(ty), shape, ne, C.byref(a), C.byref(b), C.byref(c)) == 0
    assert b"quantized" in host.thh_last_error()


def _file_with_tensor_header(path, dims, ftype=1, name=b"norm.weight", payload=b"\0" * 64):
    """A ggjt v1 file with a 3-word toy vocabulary and ONE hand-made tensor record (dims may be hostile)."""
    import struct
    with open(path, "wb") as f:
        f.write(struct.pack("<II", 0x67676A74, 1))
        f.write(struct.pack("<7i", 3, 512, 256, 8, 1, 64, 1))
        for w in (b"<unk>", b"<s>", b"</s>"):
            f.write(struct.pack("<I", len(w))); f.write(w); f.write(struct.pack("<f", 0.0))
        f.write(struct.pack("<3i", len(dims), len(name), ftype))
        f.write(struct.pack(f"<{len(dims)}i", *dims))
        f.write(name)
        f.write(b"\0" * ((-f.tell()) % 32))
        f.write(payload)


@pytest.mark.parametrize("dims,why", [((-4,), b"non-positive"), ((0, 8), b"non-positive"), ((2 ** 31 - 1, 2 ** 31 - 1), b"exceed"),
                                       ((1 << 20, 1 << 10, 1 << 10), b"exceed"), ((4096,), b"truncated tensor data")])
def test_load_llama_file_rejects_hostile_tensor_dims(host, tmp_path, dims, why):
    """negative / overflowing dimensions in the pre-scan used to give a negative record length (uncaught
    length_error in resize) or an int64 overflow; every malformed input must return false with a message instead."""
    lib = host
    lib.thh_load_file.restype = C.c_int64
    lib.thh_load_file.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.thh_last_error.restype = C.c_char_p
    path = str(tmp_path / "hostile.bin")
    _file_with_tensor_header(path, dims)
    assert lib.thh_load_file(None, path.encode(), 0) == 0          # fails in the pre-scan, before any device is needed


def test_load_llama_file_rejects_non_ggjt_without_reading_it(host, tmp_path):
    """A large file that is not ggjt v1 is rejected on its first 8 bytes (it used to be pulled into RAM prefix by prefix)."""
    import time
    lib = host
    lib.thh_load_file.restype = C.c_int64
    lib.thh_load_file.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    path = str(tmp_path / "big.bin")
    with open(path, "wb") as f:
        f.write(b"GGUF\x03\x00\x00\x00")
        f.truncate(3 << 30)                                          # sparse 3 GiB
    t0 = time.time()
    assert lib.thh_load_file(None, path.encode(), 0) == 0
    assert time.time() - t0 < 1.0
    with open(path, "r+b") as f:
        f.write(bytes.fromhex("746a6767") + b"\x02\x00\x00\x00")    # right magic, wrong version
    assert lib.thh_load_file(None, path.encode(), 0) == 0


def test_host_fp16_converters_match_reference_fixture(host):
    """The host layer's ggml_compute_fp16_to_fp32 / fp32_to_fp16 against the outputs of the REFERENCE's own functions
    (tests/golden/ref_fp16.npz, generated by tools/make_ref_fp16_golden.py from th.cpp:294-359)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_fp16.npz"))
    h = np.arange(65536, dtype=np.uint16)
    out = np.empty(65536, np.float32)
    host.thh_fp16_to_fp32(P(h), P(out), C.c_int64(h.size))
    nan = ((h & 0x7C00) == 0x7C00) & ((h & 0x3FF) != 0)
    assert (out.view(np.uint32)[~nan] == g["h2f_bits"][~nan]).all() and np.isnan(out[nan]).all()
    f = np.ascontiguousarray(g["f_in_bits"].view(np.float32))
    got = np.empty(f.size, np.uint16)
    host.thh_fp32_to_fp16(P(f), P(got), C.c_int64(f.size))
    assert (got == g["f2h"]).all()


# ------------------------------------------------------------------ reference-held pins for A22 / A21
# tests/golden/ref_host.npz was produced by the REFERENCE's own tk_llama_tokenize and llama_sample_top_p_top_k, compiled from
# /root/reference/th-llama.cpp where it lies (oracle/Makefile target _ref, tools/make_ref_host_golden.py); the fixture is data.
def _ref_host_fixture():
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_host.npz"))


def test_tokenizer_matches_reference_fixture(host):
    g = _ref_host_fixture()
    blob = g["vocab_blob"].tobytes()
    lens = np.ascontiguousarray(g["vocab_lens"]); scores = np.ascontiguousarray(g["vocab_scores"])
    tb, toff, bos = g["text_blob"].tobytes(), g["text_off"], g["text_bos"]
    ids, ioff = g["ids_blob"], g["ids_off"]
    assert len(toff) - 1 >= 100
    for i in range(len(toff) - 1):
        text = tb[toff[i]:toff[i + 1]]
        out = np.empty(len(text) + 2, np.int32)
        n = host.thh_tokenize(blob, P(lens), P(scores), lens.size, text, len(text), int(bos[i]), P(out), out.size)
        assert out[:n].tolist() == ids[ioff[i]:ioff[i + 1]].tolist(), (i, text)


def test_sampler_matches_reference_fixture(host):
    g = _ref_host_fixture()
    for c in range(g["smp_par"].shape[0]):
        k, p, t, pen = g["smp_par"][c]
        lg = np.ascontiguousarray(g["smp_logits"][c]); last = np.ascontiguousarray(g["smp_last"][c])
        want = g["smp_draws"][c]
        out = np.empty(want.size, np.int32)
        host.thh_sample(C.c_uint32(int(g["smp_seed"][c])), P(lg), lg.size, int(k), C.c_float(float(p)), C.c_float(float(t)), C.c_float(float(pen)),
                        P(last), last.size, out.size, P(out))
        assert out.tolist() == want.tolist(), (c, float(k), float(p), float(t))


def test_sampler_from_device_topk_candidates_matches_reference_fixture(host):
    """The stochastic sampler fed with the K largest raw logits only (what thk_model_logits_topk returns: value descending, ties
    by ascending id; K = top_k + distinct penalised ids + 1) instead of all n_vocab reproduces every draw of the REFERENCE's own
    llama_sample_top_p_top_k - including the parameter sets with tied logits, where the candidate path refuses (the order of equal
    values is std::partial_sort's business) and the full vector is used, exactly as th_eval does on the GPU path."""
    g = _ref_host_fixture()
    fast_total = draws_total = 0
    for c in range(g["smp_par"].shape[0]):
        k, p, t, pen = g["smp_par"][c]
        lg = np.ascontiguousarray(g["smp_logits"][c]); last = np.ascontiguousarray(g["smp_last"][c])
        want = g["smp_draws"][c]
        out = np.empty(want.size, np.int32); n_fast = C.c_int32()
        host.thh_sample_topk(C.c_uint32(int(g["smp_seed"][c])), P(lg), lg.size, int(k), C.c_float(float(p)), C.c_float(float(t)), C.c_float(float(pen)),
                             P(last), last.size, out.size, P(out), C.byref(n_fast))
        assert out.tolist() == want.tolist(), (c, float(k), float(p), float(t))
        if t > 0 and 0 < k < lg.size:
            fast_total += n_fast.value; draws_total += want.size
    assert draws_total > 0 and fast_total >= draws_total // 2, (fast_total, draws_total)   # the candidate path really carried most draws


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_sampler_from_topk_candidates_random_logits_and_penalties(host, seed):
    """Seeded logits with many penalised ids (both signs) and an engineered tie exactly at the top_k boundary: candidate path ==
    full path draw for draw."""
    rng = np.random.default_rng(seed)
    V = 5000
    lg = (rng.standard_normal(V) * 3).astype(np.float32)
    last = rng.integers(0, V, 300).astype(np.int32)
    order = np.argsort(-lg, kind="stable")
    if seed == 3:
        lg[order[40]] = lg[order[39]]                     # tie between the 40th and the 41st largest: refusal + full path
    for (k, p, t, pen) in [(40, 0.95, 0.8, 1.1), (10, 1.0, 1.3, 1.3), (100, 0.5, 0.5, 1.0)]:
        a = np.empty(64, np.int32); b = np.empty(64, np.int32); n_fast = C.c_int32()
        host.thh_sample(C.c_uint32(77 + seed), P(lg), V, k, C.c_float(p), C.c_float(t), C.c_float(pen), P(last), last.size, a.size, P(a))
        host.thh_sample_topk(C.c_uint32(77 + seed), P(lg), V, k, C.c_float(p), C.c_float(t), C.c_float(pen), P(last), last.size, b.size, P(b), C.byref(n_fast))
        assert a.tolist() == b.tolist(), (k, p, t, pen)
        if seed != 3:
            assert n_fast.value == 64


def test_python_restatements_match_reference_fixture():
    """The independent restatements used elsewhere in this file (py_tokenize, py_sample + MT19937) agree with the reference's
    own functions too, so the older restatement-based tests are anchored to the same pin."""
    g = _ref_host_fixture()
    blob, lens = g["vocab_blob"].tobytes(), g["vocab_lens"]
    words, off = [], 0
    for n in lens:
        words.append(blob[off:off + int(n)]); off += int(n)
    scores = g["vocab_scores"]
    tb, toff, bos, ids, ioff = g["text_blob"].tobytes(), g["text_off"], g["text_bos"], g["ids_blob"], g["ids_off"]
    for i in range(0, len(toff) - 1, 3):
        text = tb[toff[i]:toff[i + 1]]
        assert py_tokenize(words, scores, text, bool(bos[i])) == ids[ioff[i]:ioff[i + 1]].tolist(), (i, text)
    for c in range(g["smp_par"].shape[0]):
        k, p, t, pen = (float(v) for v in g["smp_par"][c])
        mt = MT19937(int(g["smp_seed"][c]))
        last = set(g["smp_last"][c].tolist())
        got = [py_sample(mt, g["smp_logits"][c], int(k), p, t, pen, last) for _ in range(12)]
        assert got == g["smp_draws"][c][:12].tolist(), c


def test_tensor_shape_matches_reference_fixture(host):
    """A18: TensorShape::get_total_num_elements / canonicalize as the reference's own struct computes them (th.hpp:37-77)."""
    g = _ref_host_fixture()
    for sh, want in zip(g["shp_in"], g["shp_out"]):
        out = (C.c_int64 * 5)()
        host.thh_tensor_shape_roundtrip(*[C.c_int64(int(v)) for v in sh], out)
        assert list(out) == want.tolist(), sh
