"""Writer/reader helpers for GGML 'ggjt' v1 files (format: SURVEY.md A19) used by the loader tests.
The writer emits exactly what llama.cpp's convert script emitted in May 2023: header, vocab,
then per tensor {n_dims, name_len, ftype, ne[], name, pad to a 32-byte file offset, data}."""
import struct

import numpy as np

MAGIC_GGJT, MAGIC_GGML = 0x67676A74, 0x67676D6C


def toy_vocab(n_vocab):
    """id 0 <unk>, 1 <s>, 2 </s>, 3..258 byte tokens, then a few merges, rest dummies (score -id)."""
    words = ["<unk>", "<s>", "</s>"] + [f"<0x{b:02X}>" for b in range(256)]
    scores = [0.0, 0.0, 0.0] + [-1000.0] * 256
    extra = [" ", "h", "e", "l", "o", "w", "r", "d", "he", "ll", "hel", "hell", "hello", " hello", " w", "or", "ld", "orld", " world",
             "é", "lo", " h"]
    for i, w in enumerate(extra):
        words.append(w); scores.append(-float(i + 1))
    while len(words) < n_vocab:
        words.append(f"<tok{len(words)}>"); scores.append(-float(len(words)))
    return [w.encode() for w in words[:n_vocab]], np.array(scores[:n_vocab], np.float32)


def write_ggjt(path, hp, vocab_words, vocab_scores, tensors, magic=MAGIC_GGJT, version=1):
    """hp: dict(n_vocab,n_embd,n_mult,n_head,n_layer,n_rot,ftype); tensors: list of (name, ndarray)
    with uint16 arrays meaning f16 and float32 arrays meaning f32.  Returns {name: file offset of data}."""
    offs = {}
    with open(path, "wb") as f:
        f.write(struct.pack("<II", magic, version))
        f.write(struct.pack("<7i", hp["n_vocab"], hp["n_embd"], hp["n_mult"], hp["n_head"], hp["n_layer"], hp["n_rot"], hp["ftype"]))
        for w, s in zip(vocab_words, vocab_scores):
            f.write(struct.pack("<I", len(w))); f.write(w); f.write(struct.pack("<f", float(s)))
        for name, arr in tensors:
            arr = np.ascontiguousarray(arr)
            ftype = 1 if arr.dtype == np.uint16 else 0
            dims = list(arr.shape)[::-1]            # ggml order: ne0 = columns first
            nb = name.encode()
            f.write(struct.pack("<3i", len(dims), len(nb), ftype))
            f.write(struct.pack(f"<{len(dims)}i", *dims))
            f.write(nb)
            pad = (-f.tell()) % 32
            f.write(b"\0" * pad)
            offs[name] = f.tell()
            f.write(memoryview(arr).cast("B"))
    return offs


def synthetic_model_tensors(orc, oshape):
    """(name, array) of every tensor in file order, generated one at a time (a 7B-shaped model is 13.5 GB: never all in memory)."""
    for name, dt, shp in oshape.tensor_specs():
        n = int(np.prod(shp))
        yield (name, orc.synth_f16(name, orc.TENSOR_SEED, orc.TENSOR_SIGMA, n).reshape(shp) if dt == "f16"
               else orc.synth_gain(name, orc.TENSOR_SEED, orc.TENSOR_SIGMA, n))


def write_synthetic_model(path, orc, oshape):
    words, scores = toy_vocab(oshape.n_vocab)
    hp = dict(n_vocab=oshape.n_vocab, n_embd=oshape.n_embd, n_mult=oshape.n_mult, n_head=oshape.n_head, n_layer=oshape.n_layer,
              n_rot=oshape.n_embd // oshape.n_head, ftype=1)
    return write_ggjt(path, hp, words, scores, synthetic_model_tensors(orc, oshape)), words, scores
