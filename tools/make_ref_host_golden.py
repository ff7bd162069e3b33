#!/usr/bin/env python3
"""Generate tests/golden/ref_host.npz from the REFERENCE's own tokenizer and sampler.

`make -C oracle _ref` compiles th-llama.cpp:802-907 (sample_top_k, llama_sample_top_p_top_k) and :908-1057 (TkLlamaTokenizer,
tk_llama_tokenize) straight from /root/reference — they are plain STL over LlamaVocab / the first members of LlamaModel and
need no WebGPU type — into oracle/_ref/libth_ref_host.so; this script runs them on seeded inputs and stores inputs and
outputs as plain data:
  tokenizer   vocab_blob uint8[], vocab_lens int32[V], vocab_scores float32[V]   a seeded vocabulary: specials, 256 byte tokens,
              merges over a small alphabet incl. multi-byte UTF-8 pieces, with score TIES (the bigram comparator's tie-break)
              text_blob uint8[], text_off int32[T+1], text_bos uint8[T]          the prompts
              ids_blob int32[], ids_off int32[T+1]                               tk_llama_tokenize(vocab, text, bos)
  sampler     for case c: smp_logits float32[C][N], smp_par float32[C][4] (top_k, top_p, temp, repeat_penalty), smp_seed uint32[C],
              smp_last int32[C][L], smp_draws int32[C][D]                        D consecutive llama_sample_top_p_top_k draws from ONE
                                                                                 std::mt19937(seed), as the reference seeds it
  TensorShape shp_in int64[S][4] (l,b,r,c), shp_out int64[S][5] (canonical l,b,r,c; get_total_num_elements of the shape as given), shp_str
The fixture is the reference-held pin for A21/A22 (and A18's TensorShape): tests/test_host_cpu.py checks the host layer's tokenizer and sampler
against it.  Needs /root/reference, so it runs in the build container only:  make -C oracle _ref && python tools/make_ref_host_golden.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib_path = os.path.join(ROOT, "oracle", "_ref", "libth_ref_host.so")
if not os.path.exists(lib_path):
    sys.exit("oracle/_ref/libth_ref_host.so missing: run `make -C oracle _ref` where /root/reference exists")
lib = C.CDLL(lib_path)
P = lambda a: a.ctypes.data_as(C.c_void_p)
lib.ref_tokenize.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
lib.ref_sample.argtypes = [C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p]

rng = np.random.default_rng(20230601)
# ---- vocabulary: specials, byte tokens, then merges grown from an alphabet (every merge is the concatenation of two existing
# pieces, like a BPE vocabulary), scores drawn from a SMALL set so that ties are common
words = [b"<unk>", b"<s>", b"</s>"] + [bytes([b]) for b in range(256)]
scores = [0.0, 0.0, 0.0] + [-1000.0] * 256
alphabet = [b" ", b"a", b"e", b"h", b"l", b"o", b"t", b"n", b"s", b"r", "é".encode(), "ß".encode(), "日".encode(), "本".encode(), b".", b"\n"]
pieces = list(alphabet)
seen = set(words)
while len(words) < 1500:
    a, b = pieces[int(rng.integers(len(pieces)))], pieces[int(rng.integers(len(pieces)))]
    w = a + b
    if w in seen or len(w) > 12:
        continue
    seen.add(w); words.append(w); pieces.append(w)
    scores.append(-float(rng.integers(1, 60)))          # many equal scores
vocab_blob = np.frombuffer(b"".join(words), np.uint8).copy()
vocab_lens = np.array([len(w) for w in words], np.int32)
vocab_scores = np.array(scores, np.float32)

texts = [b"", b" ", b"a", b" hello", b"hello there", " é ß 日本".encode(), b"\xff\xfe", b"\xe6\x97", b"the rest is noise.\n", b"....", b"aaaaaaaaaaaaaaaa", b" t t t t"]
for _ in range(120):
    n = int(rng.integers(1, 48))
    texts.append(b"".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), n)))
text_bos = np.array([i % 2 for i in range(len(texts))], np.uint8)
ids, ids_off, text_off = [], [0], [0]
wb = vocab_blob.tobytes()
for t, bos in zip(texts, text_bos):
    out = np.empty(len(t) + 2, np.int32)
    n = lib.ref_tokenize(wb, P(vocab_lens), P(vocab_scores), len(words), t, len(t), int(bos), P(out), out.size)
    assert n >= 0
    ids += out[:n].tolist(); ids_off.append(len(ids)); text_off.append(text_off[-1] + len(t))

# ---- sampler
N, D, L = 3000, 40, 6
cases = [(40, 0.95, 0.8, 1.1), (5, 1.0, 1.0, 1.0), (0, 0.5, 1.3, 1.3), (40, 0.95, 0.0, 1.1), (1, 0.9, 0.7, 1.1), (3000, 1.0, 2.0, 1.0), (100, 0.1, 0.5, 1.2), (40, 0.95, 0.8, 1.1)]
smp_logits = np.empty((len(cases), N), np.float32); smp_par = np.array(cases, np.float32)
smp_seed = np.array([780658349, 1, 42, 7, 123456789, 2023, 99, 780658349], np.uint32)
smp_last = np.empty((len(cases), L), np.int32); smp_draws = np.empty((len(cases), D), np.int32)
for c, (k, p, t, pen) in enumerate(cases):
    lg = (rng.standard_normal(N) * (2.0 if c != 7 else 0.01)).astype(np.float32)      # last case: nearly flat distribution
    if c == 3: lg[[17, 900]] = lg.max() + 1.0                                          # greedy with a tie: first maximum wins
    smp_logits[c] = lg
    smp_last[c] = rng.integers(0, N, L)
    out = np.empty(D, np.int32)
    lib.ref_sample(C.c_uint32(int(smp_seed[c])), P(lg), N, int(k), C.c_float(p), C.c_float(t), C.c_float(pen), P(smp_last[c]), L, D, P(out))
    smp_draws[c] = out

# ---- TensorShape (A18, th.hpp:37-77): total elements of the shape as given, canonical form, to_string
shp_in = np.array([(0, 0, 1, 4096), (1, 1, 0, 7), (0, 512, 32, 128), (0, 0, 0, 0), (1, 0, 0, 5), (2, 1, 1, 1), (0, 1, 4096, 4096), (3, 4, 5, 6),
                   (0, 0, 32000, 4096), (1, 1, 1, 1), (0, 2, 0, 9), (0, 0, 11008, 4096)], np.int64)
shp_out = np.empty((len(shp_in), 5), np.int64); shp_str = []
lib.ref_tensor_shape.argtypes = [C.c_int64] * 4 + [C.c_void_p, C.c_char_p, C.c_int]
for i, sh in enumerate(shp_in):
    buf = C.create_string_buffer(128)
    lib.ref_tensor_shape(*[int(v) for v in sh], P(shp_out[i]), buf, 128)
    shp_str.append(buf.value)
shp_str = np.array(shp_str)

dst = os.path.join(ROOT, "tests", "golden", "ref_host.npz")
np.savez_compressed(dst, vocab_blob=vocab_blob, vocab_lens=vocab_lens, vocab_scores=vocab_scores,
                    text_blob=np.frombuffer(b"".join(texts), np.uint8), text_off=np.array(text_off, np.int32), text_bos=text_bos,
                    ids_blob=np.array(ids, np.int32), ids_off=np.array(ids_off, np.int32),
                    smp_logits=smp_logits, smp_par=smp_par, smp_seed=smp_seed, smp_last=smp_last, smp_draws=smp_draws,
                    shp_in=shp_in, shp_out=shp_out, shp_str=shp_str)
print(f"wrote {dst}: {len(texts)} prompts -> {len(ids)} ids, {len(cases)} sampler cases x {D} draws, {os.path.getsize(dst)} bytes")
