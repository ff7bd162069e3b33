/*
 * thk_oracle.c — CPU restatement of TokenHawk's single-token LLaMA decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under token-hawk_amd/ (the product) may
 * include, link or dlopen this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / CPU baseline.
 *
 * PARITY UNPINNED (read before trusting):
 *   The reference (kayvr/token-hawk) executes this arithmetic as WGSL shaders
 *   inside Google Dawn, an un-vendored submodule (.gitmodules:1-3, cli/dawn is
 *   empty, no pinned commit recoverable).  All three reference TUs include
 *   <webgpu/webgpu.h>, which this image lacks, so no translation unit of the
 *   reference can be compiled here without writing a stand-in header
 *   (forbidden).  What IS compiled from the reference where it lies (line ranges
 *   that touch no WebGPU type, oracle/Makefile target _ref) are the fp16
 *   converters (A17), the sampler (A21) and the tokenizer (A22); their outputs
 *   are the fixtures tests/golden/ref_fp16.npz and ref_host.npz.  For the kernel
 *   arithmetic the reference ships no tests, golden vectors or fixtures.  This oracle is
 *   therefore a line-cited restatement of the WGSL text and host code; it is
 *   cross-checked only against independent implementations (numpy float16 for
 *   all 65,536 half patterns; a float64 numpy LLaMA forward) — see
 *   tests/test_oracle.py.  Transcendentals (exp/pow/sin/cos/sqrt) follow libm
 *   f32; WGSL leaves their precision to the backend.
 *
 * Every function cites the reference file:line it follows.  Two flavours:
 *   - "faithful" routines reproduce the shader's per-thread strip + LDS tree
 *     summation order (so a bit-level WGSL run with IEEE mul/add would match);
 *   - "fast" routines (AVX2/F16C + OpenMP, any order) are the llama.cpp-class
 *     CPU baseline timed by bench.py; tests check they agree with faithful.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -mavx2 -mfma -mf16c).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <immintrin.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))
#define WG 256 /* every reference kernel on this path uses a 256-thread workgroup */

/* ------------------------------------------------------------------------- */
/* A17: fp16 <-> fp32 (th.cpp:312-333 fp16->fp32, :335-359 fp32->fp16,        */
/*      WGSL twin th.cpp:363-394).  Bit-trick restatement.                    */
/* ------------------------------------------------------------------------- */
static inline float bits_f32(uint32_t w) { float f; memcpy(&f, &w, 4); return f; }
static inline uint32_t f32_bits(float f) { uint32_t w; memcpy(&w, &f, 4); return w; }

ORC_API float orc_fp16_to_fp32(uint16_t h) {
    const uint32_t w = (uint32_t)h << 16;
    const uint32_t sign = w & 0x80000000u;
    const uint32_t two_w = w + w;
    /* normal: shift exponent into place and rescale by 2^-112 */
    const float normalized = bits_f32((two_w >> 4) + (0xE0u << 23)) * 0x1.0p-112f;
    /* subnormal: magic-bias trick */
    const float denormalized = bits_f32((two_w >> 17) | (126u << 23)) - 0.5f;
    const uint32_t r = sign | (two_w < (1u << 27) ? f32_bits(denormalized) : f32_bits(normalized));
    return bits_f32(r);
}

ORC_API uint16_t orc_fp32_to_fp16(float f) {
    float base = (fabsf(f) * 0x1.0p+112f) * 0x1.0p-110f;
    const uint32_t w = f32_bits(f);
    const uint32_t shl1 = w + w;
    const uint32_t sign = w & 0x80000000u;
    uint32_t bias = shl1 & 0xFF000000u;
    if (bias < 0x71000000u) bias = 0x71000000u;
    base = bits_f32((bias >> 1) + 0x07800000u) + base;
    const uint32_t bits = f32_bits(base);
    const uint32_t exp_bits = (bits >> 13) & 0x00007C00u;
    const uint32_t mant = bits & 0x00000FFFu;
    const uint32_t nonsign = exp_bits + mant;
    return (uint16_t)((sign >> 16) | (shl1 > 0xFF000000u ? 0x7E00u : nonsign));
}

ORC_API void orc_fp16_to_fp32_n(const uint16_t* h, float* out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = orc_fp16_to_fp32(h[i]);
}
ORC_API void orc_fp32_to_fp16_n(const float* f, uint16_t* out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = orc_fp32_to_fp16(f[i]);
}

/* ------------------------------------------------------------------------- */
/* Synthetic weights (SURVEY.md §8d "Concrete synthetic inputs").             */
/* Counter-based and integer-only so the HIP fill kernel                      */
/* (token-hawk_amd/csrc/thk_synth.hip) produces identical bits:               */
/*   key  = fnv1a64(tensor name) ^ splitmix64(seed)                           */
/*   h    = splitmix64(key + index)                                           */
/*   s    = sum of the four 16-bit fields of h   (Irwin-Hall n=4, bell shape) */
/*   v    = (float)(s - 131070) * scale          (one f32 multiply)           */
/*   scale = sigma / 37837.2275 (std of s), f16 = RNE(v); gains = 1.0f + v    */
/* ------------------------------------------------------------------------- */
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
ORC_API uint64_t orc_synth_key(const char* name, uint64_t seed) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (const unsigned char* p = (const unsigned char*)name; *p; ++p) { h ^= *p; h *= 0x100000001B3ull; }
    return h ^ splitmix64(seed);
}
ORC_API float orc_synth_scale(float sigma) { return (float)((double)sigma / 37837.2275); }
static inline float synth_value(uint64_t key, uint64_t i, float scale) {
    const uint64_t h = splitmix64(key + i);
    const int32_t s = (int32_t)((h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48));
    return (float)(s - 131070) * scale;
}
ORC_API void orc_synth_f16(const char* name, uint64_t seed, float sigma, int64_t n, uint16_t* out) {
    const uint64_t key = orc_synth_key(name, seed);
    const float scale = orc_synth_scale(sigma);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] = _cvtss_sh(synth_value(key, (uint64_t)i, scale), _MM_FROUND_TO_NEAREST_INT);
}
ORC_API void orc_synth_gain_f32(const char* name, uint64_t seed, float sigma, int64_t n, float* out) {
    const uint64_t key = orc_synth_key(name, seed);
    const float scale = orc_synth_scale(sigma);
    for (int64_t i = 0; i < n; ++i) out[i] = 1.0f + synth_value(key, (uint64_t)i, scale);
}

/* ------------------------------------------------------------------------- */
/* Shared helper: the 256-slot LDS tree every reference kernel uses           */
/* (e.g. th.cpp:2880-2885): for stride=128..1: s[i] += s[i+stride], i<stride. */
/* ------------------------------------------------------------------------- */
static float tree_sum256(float* s) {
    for (int stride = WG / 2; stride > 0; stride /= 2)
        for (int i = 0; i < stride; ++i) s[i] = s[i] + s[i + stride];
    return s[0];
}
static float tree_max256(float* s) {
    for (int stride = WG / 2; stride > 0; stride /= 2)
        for (int i = 0; i < stride; ++i) s[i] = fmaxf(s[i], s[i + stride]);
    return s[0];
}

/* ------------------------------------------------------------------------- */
/* K1 / A6: cmdbuf_vector_mat_mul_trans (th.cpp:2839-2892; tile :2996-3006).  */
/* c[r] = sum_c a[c]*h16(b[r,c]); thread t sums the contiguous strip          */
/* [t*kTile,(t+1)*kTile) left to right, then the LDS tree.  Needs C%256==0.   */
/* ------------------------------------------------------------------------- */
ORC_API int orc_vector_mat_mul_trans(const float* a, const uint16_t* b, float* c, int64_t R, int64_t C) {
    if (C < WG || C % WG != 0) return -1;
    const int64_t kTile = C / WG;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r) {
        float sh[WG];
        const uint16_t* row = b + r * C;
        for (int t = 0; t < WG; ++t) {
            float sum = 0.0f;
            for (int64_t i = 0; i < kTile; ++i) sum = sum + a[t * kTile + i] * orc_fp16_to_fp32(row[t * kTile + i]);
            sh[t] = sum;
        }
        c[r] = tree_sum256(sh);
    }
    return 0;
}

/* fast flavour: any-order f32 accumulate, AVX2+F16C, row-parallel. */
static inline float hsum256(__m256 v) {
    __m128 lo = _mm256_castps256_ps128(v), hi = _mm256_extractf128_ps(v, 1);
    lo = _mm_add_ps(lo, hi);
    lo = _mm_add_ps(lo, _mm_movehl_ps(lo, lo));
    lo = _mm_add_ss(lo, _mm_shuffle_ps(lo, lo, 1));
    return _mm_cvtss_f32(lo);
}
static inline float dot_f16_f32(const uint16_t* w, const float* x, int64_t n) {
    __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
    int64_t i = 0;
    for (; i + 32 <= n; i += 32) {
        a0 = _mm256_fmadd_ps(_mm256_cvtph_ps(_mm_loadu_si128((const __m128i*)(w + i))), _mm256_loadu_ps(x + i), a0);
        a1 = _mm256_fmadd_ps(_mm256_cvtph_ps(_mm_loadu_si128((const __m128i*)(w + i + 8))), _mm256_loadu_ps(x + i + 8), a1);
        a2 = _mm256_fmadd_ps(_mm256_cvtph_ps(_mm_loadu_si128((const __m128i*)(w + i + 16))), _mm256_loadu_ps(x + i + 16), a2);
        a3 = _mm256_fmadd_ps(_mm256_cvtph_ps(_mm_loadu_si128((const __m128i*)(w + i + 24))), _mm256_loadu_ps(x + i + 24), a3);
    }
    float s = hsum256(_mm256_add_ps(_mm256_add_ps(a0, a1), _mm256_add_ps(a2, a3)));
    for (; i < n; ++i) s += _cvtsh_ss(w[i]) * x[i];
    return s;
}
ORC_API void orc_matvec_f16_fast(const float* a, const uint16_t* b, float* c, int64_t R, int64_t C) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r) c[r] = dot_f16_f32(b + r * C, a, C);
}

/* ------------------------------------------------------------------------- */
/* K4 / A7: cmdbuf_rms_norm (th.cpp:1153-1200).  In place, per row:           */
/* strips of N/256 squared-summed, tree, inv = 1/sqrt(s/N + 1e-6), x *= inv.  */
/* ------------------------------------------------------------------------- */
ORC_API int orc_rms_norm(float* x, int64_t rows, int64_t N) {
    if (N < WG || N % WG != 0) return -1;
    const int64_t per = N / WG;
    for (int64_t r = 0; r < rows; ++r) {
        float* row = x + r * N;
        float sh[WG];
        for (int t = 0; t < WG; ++t) {
            float sum = 0.0f;
            for (int64_t i = 0; i < per; ++i) sum = sum + row[t * per + i] * row[t * per + i];
            sh[t] = sum;
        }
        const float inv = 1.0f / sqrtf(tree_sum256(sh) / (float)N + 1e-6f);
        for (int64_t i = 0; i < N; ++i) row[i] = row[i] * inv;
    }
    return 0;
}

/* K5 / A8: cmdbuf_row_element_multiply (th.cpp:1298-1315): x[r,c] *= w[c]. */
ORC_API void orc_row_element_multiply(float* x, const float* w, int64_t rows, int64_t N) {
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < N; ++c) x[r * N + c] = x[r * N + c] * w[c];
}

/* ------------------------------------------------------------------------- */
/* K6 / A9: cmdbuf_RoPE (th.cpp:1452-1492; constants :1520-1526; host shapes  */
/* th-llama.cpp:317-322).  x viewed [n_tok, H, D]; for even j:                */
/*   theta = pow(10000, -j/D); a = f32(n_past + tok) * theta;                 */
/*   (x0,x1) -> (x0 cos a - x1 sin a, x0 sin a + x1 cos a).                   */
/* ------------------------------------------------------------------------- */
ORC_API void orc_rope_angles(int64_t D, int64_t pos, float* cos_out, float* sin_out) {
    for (int64_t j = 0; j < D; j += 2) {
        const float theta = powf(10000.0f, (-(float)j) / (float)D);
        const float p = (float)pos;
        cos_out[j / 2] = cosf(p * theta);
        sin_out[j / 2] = sinf(p * theta);
    }
}
ORC_API void orc_rope(float* x, int64_t n_tok, int64_t H, int64_t D, int64_t n_past) {
    float cs[D / 2], sn[D / 2];
    for (int64_t t = 0; t < n_tok; ++t) {
        orc_rope_angles(D, n_past + t, cs, sn);
        for (int64_t h = 0; h < H; ++h) {
            float* v = x + (t * H + h) * D;
            for (int64_t j = 0; j < D; j += 2) {
                const float x0 = v[j], x1 = v[j + 1];
                v[j] = x0 * cs[j / 2] - x1 * sn[j / 2];
                v[j + 1] = x0 * sn[j / 2] + x1 * cs[j / 2];
            }
        }
    }
}

/* K8 / A11: cmdbuf_transpose, zy mode (th.cpp:863-912): [B,M,N] -> [M,B,N]. */
ORC_API void orc_transpose_zy(const float* a, float* c, int64_t B, int64_t M, int64_t N) {
    for (int64_t z = 0; z < B; ++z)
        for (int64_t y = 0; y < M; ++y)
            memcpy(c + (y * B + z) * N, a + (z * M + y) * N, (size_t)N * sizeof(float));
}

/* ------------------------------------------------------------------------- */
/* K9 / A12: cmdbuf_mat_mul (th.cpp:396-539), f32 B operand, 8x8 workgroup,   */
/* 1x1 register tile: K walked in chunks of 8 (zero-padded); each chunk is    */
/* summed left-to-right from 0 and added to the running total; optional       */
/* post-scale (th.cpp:527-529).  C[z] = A[z] * (transposeB ? B[z]^T : B[z]).  */
/* A [Bz,M,K]; B [Bz,K,N] (or [Bz,N,K] when transposed); C [Bz,M,N].          */
/* ------------------------------------------------------------------------- */
ORC_API void orc_mat_mul(const float* A, const float* B, float* C, int64_t Bz, int64_t M, int64_t K, int64_t N,
                         int transposeB, int do_scale, float scale) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t z = 0; z < Bz; ++z)
        for (int64_t m = 0; m < M; ++m)
            for (int64_t n = 0; n < N; ++n) {
                const float* a = A + (z * M + m) * K;
                const float* b = B + z * K * N;
                float total = 0.0f;
                for (int64_t k0 = 0; k0 < K; k0 += 8) {
                    float sum = 0.0f;
                    for (int64_t k = k0; k < k0 + 8; ++k) {
                        const float av = k < K ? a[k] : 0.0f;
                        const float bv = k < K ? (transposeB ? b[n * K + k] : b[k * N + n]) : 0.0f;
                        sum = sum + av * bv;
                    }
                    total = total + sum;
                }
                if (do_scale) total = total * scale;
                C[(z * M + m) * N + n] = total;
            }
}

/* ------------------------------------------------------------------------- */
/* K10 / A13: cmdbuf_row_softmax (th.cpp:1865-1961).  Per row of N: thread t  */
/* owns the strip [t*per,(t+1)*per), per = ceil(N/256); max with -1e14 init,  */
/* tree; exp(x-max) summed per strip, tree; divide.                           */
/* ------------------------------------------------------------------------- */
ORC_API void orc_row_softmax(float* a, int64_t rows, int64_t N) {
    const int64_t per = (N + WG - 1) / WG;
    for (int64_t r = 0; r < rows; ++r) {
        float* row = a + r * N;
        float sh[WG];
        for (int t = 0; t < WG; ++t) {
            float mx = -1e14f;
            for (int64_t i = 0; i < per; ++i) if (t * per + i < N) mx = fmaxf(mx, row[t * per + i]);
            sh[t] = mx;
        }
        const float row_max = tree_max256(sh);
        for (int t = 0; t < WG; ++t) {
            float sum = 0.0f;
            for (int64_t i = 0; i < per; ++i)
                if (t * per + i < N) {
                    const float e = expf(row[t * per + i] - row_max);
                    row[t * per + i] = e;
                    sum = sum + e;
                }
            sh[t] = sum;
        }
        const float denom = tree_sum256(sh);
        for (int64_t i = 0; i < N; ++i) row[i] = row[i] / denom;
    }
}

/* K11 (th.cpp:2121-2149), K12 (:2680-2709), K13 (:2498-2526) / A14. */
ORC_API void orc_addition(const float* a, const float* b, float* c, int64_t n) {
    for (int64_t i = 0; i < n; ++i) c[i] = a[i] + b[i];
}
ORC_API void orc_silu(float* a, int64_t n) {
    for (int64_t i = 0; i < n; ++i) { const float v = a[i]; a[i] = v / (1.0f + expf(-v)); }
}
ORC_API void orc_element_mult_in_place(float* a, const float* b, int64_t n) {
    for (int64_t i = 0; i < n; ++i) a[i] = a[i] * b[i];
}

/* ------------------------------------------------------------------------- */
/* K2 / A15: cmdbuf_vector_multi_mat_mul_split_trans (th.cpp:3516-3586):      */
/* lm-head as two half-K mat-vecs over the two physically separate halves     */
/* [V, E/2] made by the loader (th-llama-loader.cpp:197-242).  We index the   */
/* unsplit row-major matrix: half s covers columns [s*E/2,(s+1)*E/2).         */
/* Each half: strips of (E/2)/256, tree.  Needs (E/2)%256==0.                 */
/* ------------------------------------------------------------------------- */
ORC_API int orc_lmhead_split(const float* x, const uint16_t* W, float* out, float* scratch, int64_t V, int64_t E) {
    const int64_t Ch = E / 2;
    if (Ch < WG || Ch % WG != 0) return -1;
    const int64_t kTile = Ch / WG;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < V; ++r)
        for (int s = 0; s < 2; ++s) {
            float sh[WG];
            const uint16_t* row = W + r * E + s * Ch;
            const float* xa = x + s * Ch;
            for (int t = 0; t < WG; ++t) {
                float sum = 0.0f;
                for (int64_t i = 0; i < kTile; ++i) sum = sum + xa[t * kTile + i] * orc_fp16_to_fp32(row[t * kTile + i]);
                sh[t] = sum;
            }
            (s == 0 ? out : scratch)[r] = tree_sum256(sh);
        }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* K3 / A16: cmdbuf_vector_reduce (th.cpp:3914-3945; constants :3990-4018),   */
/* called with numSplits=8 (th-llama.cpp:262).  splitSize = C/8,              */
/* kTile = max(1, (C/8)/256); workgroup w, thread t covers                    */
/* [w*splitSize + t*kTile, +kTile).  faithful=1 reproduces defect Q1 (only    */
/* 256*kTile of each splitSize entries are summed); faithful=0 sums all.      */
/* ------------------------------------------------------------------------- */
ORC_API int64_t orc_q1_covered(int64_t C, int64_t i) { /* 1 if index i is summed by the faithful combine */
    const int64_t split = C / 8;
    int64_t kTile = split / WG; if (kTile == 0) kTile = 1;
    if (i < 0 || i >= C) return 0;
    for (int64_t w = 0; w < 8; ++w) { const int64_t lo = w * split; if (i >= lo && i < lo + WG * kTile) return 1; }
    return 0;
}
ORC_API void orc_vector_reduce(float* a, const float* b, int64_t C, int faithful) {
    if (!faithful) { for (int64_t i = 0; i < C; ++i) a[i] = a[i] + b[i]; return; }
    const int64_t split = C / 8;
    int64_t kTile = split / WG; if (kTile == 0) kTile = 1;
    /* NOTE: when 256*kTile > split, ranges of adjacent workgroups overlap and the shader
     * races; this only happens for C < 2048 which no supported model has.  We add once. */
    uint8_t* done = (uint8_t*)calloc((size_t)C, 1);
    for (int64_t w = 0; w < 8; ++w)
        for (int t = 0; t < WG; ++t)
            for (int64_t i = 0; i < kTile; ++i) {
                const int64_t idx = w * split + t * kTile + i;
                if (idx < C && !done[idx]) { a[idx] = a[idx] + b[idx]; done[idx] = 1; }
            }
    free(done);
}

/* A21 greedy branch: llama_sample_top_p_top_k with temp<=0 (th-llama.cpp:826-838):
 * first index attaining the maximum (strict '>' scan). */
ORC_API int32_t orc_greedy(const float* logits, int64_t n) {
    float best = logits[0]; int32_t id = 0;
    for (int64_t i = 1; i < n; ++i) if (logits[i] > best) { best = logits[i]; id = (int32_t)i; }
    return id;
}

/* ========================================================================= */
/* Model: LlamaModel / LlamaLayer (th-llama.hpp:37-55, :100-179), buffers as  */
/* allocated by post_load_init_model (th-llama-loader.cpp:330-435).           */
/* ========================================================================= */
typedef struct {
    int32_t n_vocab, n_embd, n_mult, n_head, n_layer, n_ctx;
} orc_hparams;

typedef struct {
    float* attention_norm; uint16_t *wq, *wk, *wv, *wo;
    float* ffn_norm; uint16_t *w1, *w2, *w3;
    float **key_cache, **value_cache; /* [n_seq] each f32 [n_ctx, H, D] */
} orc_layer;

typedef struct orc_model {
    orc_hparams hp; int32_t n_ff, n_seq;
    uint16_t* tok_embeddings; /* f16 [V,E]; the reference converts it to f32 at load (loader :185-195) */
    float* norm; uint16_t* output; /* [V,E] */
    orc_layer* layers;
    /* working buffers: inp[0..6], ffWorking[0..1], working K/V, inp5 scores */
    float *inp[7], *ff[2], *wk_cache, *wv_cache, *out, *out_scratch;
} orc_model;

ORC_API int32_t orc_n_ff(int32_t n_embd, int32_t n_mult) { /* th-llama-loader.cpp:349 */
    return ((2 * (4 * n_embd) / 3 + n_mult - 1) / n_mult) * n_mult;
}

ORC_API orc_model* orc_model_create(const orc_hparams* hp, int32_t n_seq) {
    orc_model* m = (orc_model*)calloc(1, sizeof(orc_model));
    m->hp = *hp; m->n_seq = n_seq;
    m->n_ff = orc_n_ff(hp->n_embd, hp->n_mult);
    const int64_t E = hp->n_embd, F = m->n_ff, V = hp->n_vocab, L = hp->n_layer, T = hp->n_ctx;
    m->tok_embeddings = (uint16_t*)malloc((size_t)(V * E) * 2);
    m->output = (uint16_t*)malloc((size_t)(V * E) * 2);
    m->norm = (float*)malloc((size_t)E * 4);
    m->layers = (orc_layer*)calloc((size_t)L, sizeof(orc_layer));
    for (int64_t l = 0; l < L; ++l) {
        orc_layer* y = &m->layers[l];
        y->attention_norm = (float*)malloc((size_t)E * 4);
        y->ffn_norm = (float*)malloc((size_t)E * 4);
        y->wq = (uint16_t*)malloc((size_t)(E * E) * 2); y->wk = (uint16_t*)malloc((size_t)(E * E) * 2);
        y->wv = (uint16_t*)malloc((size_t)(E * E) * 2); y->wo = (uint16_t*)malloc((size_t)(E * E) * 2);
        y->w1 = (uint16_t*)malloc((size_t)(F * E) * 2); y->w3 = (uint16_t*)malloc((size_t)(F * E) * 2);
        y->w2 = (uint16_t*)malloc((size_t)(E * F) * 2);
        y->key_cache = (float**)calloc((size_t)n_seq, sizeof(float*));
        y->value_cache = (float**)calloc((size_t)n_seq, sizeof(float*));
        for (int s = 0; s < n_seq; ++s) {
            y->key_cache[s] = (float*)calloc((size_t)(T * E), 4);
            y->value_cache[s] = (float*)calloc((size_t)(T * E), 4);
        }
    }
    for (int i = 0; i < 7; ++i) m->inp[i] = (float*)calloc((size_t)(E > T * hp->n_head ? E : T * hp->n_head), 4);
    m->ff[0] = (float*)calloc((size_t)F, 4); m->ff[1] = (float*)calloc((size_t)F, 4);
    m->wk_cache = (float*)calloc((size_t)(T * E), 4); m->wv_cache = (float*)calloc((size_t)(T * E), 4);
    m->out = (float*)calloc((size_t)V, 4); m->out_scratch = (float*)calloc((size_t)V, 4);
    return m;
}

ORC_API void orc_model_destroy(orc_model* m) {
    if (!m) return;
    for (int64_t l = 0; l < m->hp.n_layer; ++l) {
        orc_layer* y = &m->layers[l];
        free(y->attention_norm); free(y->ffn_norm); free(y->wq); free(y->wk); free(y->wv); free(y->wo);
        free(y->w1); free(y->w2); free(y->w3);
        for (int s = 0; s < m->n_seq; ++s) { free(y->key_cache[s]); free(y->value_cache[s]); }
        free(y->key_cache); free(y->value_cache);
    }
    free(m->layers); free(m->tok_embeddings); free(m->output); free(m->norm);
    for (int i = 0; i < 7; ++i) free(m->inp[i]);
    free(m->ff[0]); free(m->ff[1]); free(m->wk_cache); free(m->wv_cache); free(m->out); free(m->out_scratch);
    free(m);
}

/* Tensor names as in the ggjt file (th-llama-loader.cpp:410-426). Returns the
 * destination pointer and element count/type for `name`, or NULL. */
static void* tensor_slot(orc_model* m, const char* name, int64_t* n, int* is_f16) {
    const int64_t E = m->hp.n_embd, F = m->n_ff, V = m->hp.n_vocab;
    if (!strcmp(name, "tok_embeddings.weight")) { *n = V * E; *is_f16 = 1; return m->tok_embeddings; }
    if (!strcmp(name, "norm.weight")) { *n = E; *is_f16 = 0; return m->norm; }
    if (!strcmp(name, "output.weight")) { *n = V * E; *is_f16 = 1; return m->output; }
    int l = -1; char rest[64];
    if (sscanf(name, "layers.%d.%63s", &l, rest) == 2 && l >= 0 && l < m->hp.n_layer) {
        orc_layer* y = &m->layers[l];
        if (!strcmp(rest, "attention_norm.weight")) { *n = E; *is_f16 = 0; return y->attention_norm; }
        if (!strcmp(rest, "ffn_norm.weight")) { *n = E; *is_f16 = 0; return y->ffn_norm; }
        if (!strcmp(rest, "attention.wq.weight")) { *n = E * E; *is_f16 = 1; return y->wq; }
        if (!strcmp(rest, "attention.wk.weight")) { *n = E * E; *is_f16 = 1; return y->wk; }
        if (!strcmp(rest, "attention.wv.weight")) { *n = E * E; *is_f16 = 1; return y->wv; }
        if (!strcmp(rest, "attention.wo.weight")) { *n = E * E; *is_f16 = 1; return y->wo; }
        if (!strcmp(rest, "feed_forward.w1.weight")) { *n = F * E; *is_f16 = 1; return y->w1; }
        if (!strcmp(rest, "feed_forward.w2.weight")) { *n = E * F; *is_f16 = 1; return y->w2; }
        if (!strcmp(rest, "feed_forward.w3.weight")) { *n = F * E; *is_f16 = 1; return y->w3; }
    }
    return NULL;
}

ORC_API int orc_model_set_tensor(orc_model* m, const char* name, const void* data, int64_t n_elements) {
    int64_t n; int f16; void* dst = tensor_slot(m, name, &n, &f16);
    if (!dst || n != n_elements) return -1;
    memcpy(dst, data, (size_t)n * (f16 ? 2 : 4));
    return 0;
}
ORC_API int orc_model_get_tensor(orc_model* m, const char* name, void* out, int64_t n_elements) {
    int64_t n; int f16; void* src = tensor_slot(m, name, &n, &f16);
    if (!src || n != n_elements) return -1;
    memcpy(out, src, (size_t)n * (f16 ? 2 : 4));
    return 0;
}

/* Fill every tensor of layers [l0,l1) (+embeddings/head when asked) with the
 * synthetic generator; names are the ggjt names so any shard agrees. */
ORC_API void orc_model_fill_synthetic(orc_model* m, uint64_t seed, float sigma, int l0, int l1, int with_embed, int with_head) {
    const int64_t E = m->hp.n_embd, F = m->n_ff, V = m->hp.n_vocab;
    char nm[96];
    if (with_embed) orc_synth_f16("tok_embeddings.weight", seed, sigma, V * E, m->tok_embeddings);
    if (with_head) {
        orc_synth_gain_f32("norm.weight", seed, sigma, E, m->norm);
        orc_synth_f16("output.weight", seed, sigma, V * E, m->output);
    }
    for (int l = l0; l < l1; ++l) {
        orc_layer* y = &m->layers[l];
#define NM(s) (snprintf(nm, sizeof nm, "layers.%d." s, l), nm)
        orc_synth_gain_f32(NM("attention_norm.weight"), seed, sigma, E, y->attention_norm);
        orc_synth_gain_f32(NM("ffn_norm.weight"), seed, sigma, E, y->ffn_norm);
        orc_synth_f16(NM("attention.wq.weight"), seed, sigma, E * E, y->wq);
        orc_synth_f16(NM("attention.wk.weight"), seed, sigma, E * E, y->wk);
        orc_synth_f16(NM("attention.wv.weight"), seed, sigma, E * E, y->wv);
        orc_synth_f16(NM("attention.wo.weight"), seed, sigma, E * E, y->wo);
        orc_synth_f16(NM("feed_forward.w1.weight"), seed, sigma, F * E, y->w1);
        orc_synth_f16(NM("feed_forward.w2.weight"), seed, sigma, E * F, y->w2);
        orc_synth_f16(NM("feed_forward.w3.weight"), seed, sigma, F * E, y->w3);
#undef NM
    }
}

ORC_API void orc_model_reset_kv(orc_model* m, int seq) {
    const size_t n = (size_t)m->hp.n_ctx * m->hp.n_embd * 4;
    for (int l = 0; l < m->hp.n_layer; ++l) { memset(m->layers[l].key_cache[seq], 0, n); memset(m->layers[l].value_cache[seq], 0, n); }
}
ORC_API float* orc_model_kv_ptr(orc_model* m, int layer, int seq, int which) {
    return which == 0 ? m->layers[layer].key_cache[seq] : m->layers[layer].value_cache[seq];
}

/* Embedding fetch (th-llama-loader.cpp:185-195 + th-llama.cpp:577-584):
 * x = res = h16(tok_embeddings[token,:]) written to inp0 and inp6. */
ORC_API void orc_embed(const orc_model* m, int32_t token, float* x) {
    const int64_t E = m->hp.n_embd;
    for (int64_t i = 0; i < E; ++i) x[i] = orc_fp16_to_fp32(m->tok_embeddings[(int64_t)token * E + i]);
}

typedef void (*matvec_fn)(const float*, const uint16_t*, float*, int64_t, int64_t);
static void mv_faithful(const float* a, const uint16_t* b, float* c, int64_t R, int64_t C) { orc_vector_mat_mul_trans(a, b, c, R, C); }

/* ------------------------------------------------------------------------- */
/* One transformer layer, single token: build_layer_cmdbuf step order         */
/* (th-llama.cpp:270-452).  `x` is inp0 on entry        */
/* (== residual inp6) and holds the layer output on exit.                     */
/* faithful=1: reference kernels incl. the transposes and K9 tile order.      */
/* faithful=0: fast flavour (direct cache indexing, any-order sums).          */
/* ------------------------------------------------------------------------- */
static void layer_forward(orc_model* m, int l, int seq, float* x, int n_past, int faithful, int kv_f16) {
    const int64_t E = m->hp.n_embd, H = m->hp.n_head, D = E / H, F = m->n_ff, T = n_past + 1;
    orc_layer* y = &m->layers[l];
    matvec_fn mv = faithful ? mv_faithful : orc_matvec_f16_fast;
    float *inp0 = m->inp[0], *q = m->inp[1], *k = m->inp[2], *v = m->inp[3], *qT = m->inp[4], *S = m->inp[5], *res = m->inp[6];
    memcpy(inp0, x, (size_t)E * 4); memcpy(res, x, (size_t)E * 4);
    /* 1: rms_norm + gain (:299-300) */
    orc_rms_norm(inp0, 1, E); orc_row_element_multiply(inp0, y->attention_norm, 1, E);
    /* 2: q,k,v projections (:303-306) */
    mv(inp0, y->wq, q, E, E); mv(inp0, y->wk, k, E, E); mv(inp0, y->wv, v, E, E);
    /* 3: RoPE on q,k viewed [1,H,D] (:317-322) */
    orc_rope(q, 1, H, D, n_past); orc_rope(k, 1, H, D, n_past);
    /* 4: append to caches at row n_past (:332-339) */
    float *Kc = y->key_cache[seq], *Vc = y->value_cache[seq];
    if (kv_f16) {   /* optional f16 KV cache of the build: k, v are rounded to binary16 (RNE) as they are appended */
        for (int64_t i = 0; i < E; ++i) { k[i] = orc_fp16_to_fp32(orc_fp32_to_fp16(k[i])); v[i] = orc_fp16_to_fp32(orc_fp32_to_fp16(v[i])); }
    }
    memcpy(Kc + (int64_t)n_past * E, k, (size_t)E * 4); memcpy(Vc + (int64_t)n_past * E, v, (size_t)E * 4);
    float* o = m->inp[2]; /* keyBuf is reused for the attention output (:380) */
    if (faithful) {
        /* 5: transposes [T,H,D]->[H,T,D]; q [1,H,D]->[H,1,D] (:341-355) */
        orc_transpose_zy(Kc, m->wk_cache, T, H, D); orc_transpose_zy(Vc, m->wv_cache, T, H, D);
        orc_transpose_zy(q, qT, 1, H, D);
        /* 6: S[h,0,t] = (q_h . k_{h,t}) / sqrt(D)  (:361-365; scale th-llama.cpp:518) */
        orc_mat_mul(qT, m->wk_cache, S, H, 1, D, T, 1, 1, 1.0f / sqrtf((float)D));
        /* 7: softmax rows (:373) */
        orc_row_softmax(S, H, T);
        /* 8: o[h,0,:] = P[h,:] . V[h,:,:], scale 1.0 (:380, uniforms :538-550) */
        orc_mat_mul(S, m->wv_cache, o, H, 1, T, D, 0, 1, 1.0f);
        /* 9: [H,1,D] -> [1,H,D] (:396-397) is the identity relabel for one token */
        orc_transpose_zy(o, v, H, 1, D);
    } else {
        const float scale = 1.0f / sqrtf((float)D);
#pragma omp parallel for schedule(static)
        for (int64_t h = 0; h < H; ++h) {
            float* s = S + h * T; float mx = -1e14f;
            for (int64_t t = 0; t < T; ++t) {
                float acc = 0.0f; const float* kk = Kc + t * E + h * D;
                for (int64_t d = 0; d < D; ++d) acc += q[h * D + d] * kk[d];
                s[t] = acc * scale; mx = fmaxf(mx, s[t]);
            }
            float den = 0.0f;
            for (int64_t t = 0; t < T; ++t) { s[t] = expf(s[t] - mx); den += s[t]; }
            float* oo = v + h * D;
            for (int64_t d = 0; d < D; ++d) oo[d] = 0.0f;
            for (int64_t t = 0; t < T; ++t) { const float p = s[t] / den; const float* vv = Vc + t * E + h * D; for (int64_t d = 0; d < D; ++d) oo[d] += p * vv[d]; }
        }
    }
    /* 10: wo (:401-402); 11: + residual, keep copy (:409-413) */
    mv(v, y->wo, q, E, E);
    float* x2 = m->inp[2]; orc_addition(q, res, x2, E);
    float* res2 = m->inp[3]; memcpy(res2, x2, (size_t)E * 4);
    /* 12: ffn norm (:415-416) */
    orc_rms_norm(x2, 1, E); orc_row_element_multiply(x2, y->ffn_norm, 1, E);
    /* 13: w1, w3 (:422-424); 14: silu, gate (:436-438); 15: w2 (:440-441) */
    mv(x2, y->w1, m->ff[0], F, E); mv(x2, y->w3, m->ff[1], F, E);
    orc_silu(m->ff[0], F); orc_element_mult_in_place(m->ff[0], m->ff[1], F);
    mv(m->ff[0], y->w2, x2, E, F);
    /* 16: x = res2 + d (:447-451) */
    orc_addition(res2, x2, x, E);
}

/* Final stage: build_final_compute_cmdbuf (th-llama.cpp:240-268).
 * lm_faithful=1 reproduces Q1; 0 = full sum. */
static void head_forward(orc_model* m, const float* x, float* logits, int faithful, int lm_faithful) {
    const int64_t E = m->hp.n_embd, V = m->hp.n_vocab;
    float* n = m->inp[0]; memcpy(n, x, (size_t)E * 4);
    orc_rms_norm(n, 1, E); orc_row_element_multiply(n, m->norm, 1, E);
    if (faithful) {
        orc_lmhead_split(n, m->output, m->out, m->out_scratch, V, E);
        orc_vector_reduce(m->out, m->out_scratch, V, lm_faithful);
        memcpy(logits, m->out, (size_t)V * 4);
    } else {
        const int64_t Ch = E / 2;
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < V; ++r) {
            const float p0 = dot_f16_f32(m->output + r * E, n, Ch);
            const float p1 = dot_f16_f32(m->output + r * E + Ch, n + Ch, Ch);
            logits[r] = (lm_faithful && !orc_q1_covered(V, r)) ? p0 : p0 + p1;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* th_eval_gpu restated for n_tokens=1 (th-llama.cpp:464-660), generalised to */
/* a layer range so pipeline stages can be checked:                           */
/*   token >= 0 : x = embedding(token); else x = hidden_inout on entry        */
/*   runs layers [l0,l1); writes x back to hidden_inout (if non-NULL)         */
/*   logits != NULL : final norm + lm-head into logits[V]                     */
/* flags: bit0 faithful summation order, bit1 lm-head Q1-faithful combine,     */
/*        bit2 K/V rounded to f16 at the append (the build's f16-KV option).  */
/* ------------------------------------------------------------------------- */
ORC_API int orc_model_eval(orc_model* m, int seq, int32_t token, int n_past, int l0, int l1,
                           float* hidden_inout, float* logits, int flags) {
    const int64_t E = m->hp.n_embd;
    if (n_past < 0 || n_past >= m->hp.n_ctx || seq < 0 || seq >= m->n_seq) return -1;
    float* x = (float*)malloc((size_t)E * 4);
    if (token >= 0) orc_embed(m, token, x); else if (hidden_inout) memcpy(x, hidden_inout, (size_t)E * 4); else { free(x); return -2; }
    for (int l = l0; l < l1; ++l) layer_forward(m, l, seq, x, n_past, flags & 1, (flags >> 2) & 1);
    if (hidden_inout) memcpy(hidden_inout, x, (size_t)E * 4);
    if (logits) head_forward(m, x, logits, flags & 1, (flags >> 1) & 1);
    free(x);
    return 0;
}

/* Bounded CPU-baseline sample for bench.py: time `steps` decode steps at a
 * fixed n_past over layers [0,n_layers_sample) + the head; returns seconds for
 * layers and head separately so the caller can extrapolate to n_layer. */
ORC_API int orc_model_time_decode(orc_model* m, int n_past, int n_layers_sample, int steps, double* sec_layers, double* sec_head) {
#ifdef _OPENMP
    const int64_t E = m->hp.n_embd, V = m->hp.n_vocab;
    float* x = (float*)malloc((size_t)E * 4); float* lg = (float*)malloc((size_t)V * 4);
    double tl = 0, th = 0;
    for (int s = 0; s < steps; ++s) {
        orc_embed(m, 1 + s, x);
        double t0 = omp_get_wtime();
        for (int l = 0; l < n_layers_sample; ++l) layer_forward(m, l, 0, x, n_past, 0, 0);
        double t1 = omp_get_wtime();
        head_forward(m, x, lg, 0, 0);
        double t2 = omp_get_wtime();
        tl += t1 - t0; th += t2 - t1;
    }
    *sec_layers = tl; *sec_head = th; free(x); free(lg);
    return 0;
#else
    (void)m; (void)n_past; (void)n_layers_sample; (void)steps; (void)sec_layers; (void)sec_head; return -1;
#endif
}
ORC_API void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
