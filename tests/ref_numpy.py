"""Independent float64 numpy LLaMA forward (written from the architecture, not from the
oracle) used to cross-check oracle/thk_oracle.c.  Interleaved-pair RoPE, RMSNorm eps 1e-6,
SwiGLU, f32 KV semantics ignored (everything float64)."""
import numpy as np


def f16_to_f64(u16):
    return np.asarray(u16, np.uint16).view(np.float16).astype(np.float64)


class RefModel:
    def __init__(self, shape, tensors):
        self.s = shape
        self.t = {k: (f16_to_f64(v) if v.dtype == np.uint16 else v.astype(np.float64)) for k, v in tensors.items()}
        E = shape.n_embd
        self.K = [np.zeros((shape.n_ctx, E)) for _ in range(shape.n_layer)]
        self.V = [np.zeros((shape.n_ctx, E)) for _ in range(shape.n_layer)]

    @staticmethod
    def rms(x, g):
        return x / np.sqrt(np.mean(x * x) + 1e-6) * g

    def rope(self, v, pos):
        H, D = self.s.n_head, self.s.n_embd // self.s.n_head
        v = v.reshape(H, D).copy()
        j = np.arange(0, D, 2)
        ang = pos * (10000.0 ** (-j / D))
        c, s = np.cos(ang), np.sin(ang)
        x0, x1 = v[:, 0::2].copy(), v[:, 1::2].copy()
        v[:, 0::2] = x0 * c - x1 * s
        v[:, 1::2] = x0 * s + x1 * c
        return v.reshape(-1)

    def eval(self, token, n_past, q1_faithful=False):
        s, t = self.s, self.t
        E, H = s.n_embd, s.n_head
        D = E // H
        x = t["tok_embeddings.weight"][token].copy()
        for l in range(s.n_layer):
            p = f"layers.{l}."
            n = self.rms(x, t[p + "attention_norm.weight"])
            q = self.rope(t[p + "attention.wq.weight"] @ n, n_past)
            k = self.rope(t[p + "attention.wk.weight"] @ n, n_past)
            v = t[p + "attention.wv.weight"] @ n
            self.K[l][n_past] = k
            self.V[l][n_past] = v
            T = n_past + 1
            o = np.zeros(E)
            for h in range(H):
                sl = slice(h * D, (h + 1) * D)
                sc = self.K[l][:T, sl] @ q[sl] / np.sqrt(D)
                pr = np.exp(sc - sc.max())
                pr /= pr.sum()
                o[sl] = pr @ self.V[l][:T, sl]
            x = x + t[p + "attention.wo.weight"] @ o
            n = self.rms(x, t[p + "ffn_norm.weight"])
            a = t[p + "feed_forward.w1.weight"] @ n
            b = t[p + "feed_forward.w3.weight"] @ n
            x = x + t[p + "feed_forward.w2.weight"] @ ((a / (1.0 + np.exp(-a))) * b)
        n = self.rms(x, t["norm.weight"])
        W = t["output.weight"]
        logits = W @ n
        if q1_faithful:
            V = s.n_vocab
            split = V // 8
            ktile = max(1, split // 256)
            r = np.arange(V)
            skipped = (r % split) >= 256 * ktile
            half = W[:, : E // 2] @ n[: E // 2]
            logits = np.where(skipped, half, logits)
        return logits
