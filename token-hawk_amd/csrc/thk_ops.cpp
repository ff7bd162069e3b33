// thk_ops.cpp — the operator level of the C-ABI: one entry point per reference kernel (the 16 cmdbuf_* encoders and
// their WGSL, th.cpp:396-4351), explicit dimensions instead of baked shader constants.  Used by the parity tests per op;
// the model level (thk_model.cpp) launches the fused forms directly.
#include "thk_internal.hpp"
#include <sched.h>
#include <time.h>

// ---------------------------------------------------------------- operators
static int gemv_simple(thk_ctx* ctx, int pro, int epi, const char* var_name, const char* bpc_name, GemvArgs& a, int rows) {
    int nru = (int)tun(ctx, var_name);
    if (nru < 0) nru = auto_geometry(var_name + 13 /* past "gemv_variant_" */, a.C == 5120 ? 5120 : 4096).var;
    const int NR = gemv_rows_per_group(a.C, epi, nru);
    a.n_groups = (rows + NR - 1) / NR;
    const int grid = grid_for(ctx, bpc_name, a.n_groups, a.C == 5120 ? 5120 : 4096);
    HIPCHK(ctx, launch_gemv(pro, epi, nru, a, grid, true, ctx->stream));
    return grid;
}

extern "C" int thk_matvec_f16(thk_ctx* ctx, const void* W, int64_t R, int64_t C, const float* x, float* y) {
    if (!ctx) return THK_ERR_INVALID;
    REQUIRE(ctx, W && x && y && R > 0, "thk_matvec_f16: null pointer or empty matrix");
    REQUIRE(ctx, C >= 256 && C % 256 == 0, "thk_matvec_f16: C=%lld must be a multiple of 256 (th.cpp:2996-3006)", (long long)C);
    REQUIRE(ctx, C <= 32768 && R <= 0x7FFFFFFF, "thk_matvec_f16: shape too large");
    GemvArgs a{}; a.W[0] = (const uint16_t*)W; a.R = (int)R; a.C = (int)C; a.x = x; a.y = y;
    const int rc = gemv_simple(ctx, GEMV_PRO_COPY, GEMV_EPI_STORE, C > 8192 ? "gemv_variant_w2" : "gemv_variant_wo", "gemv_blocks_per_cu", a, (int)R);
    return rc < 0 ? rc : THK_OK;
}
extern "C" int thk_rms_norm(thk_ctx* ctx, float* x, int64_t rows, int64_t N) {
    if (!ctx) return THK_ERR_INVALID;
    REQUIRE(ctx, x && rows > 0 && N > 0, "thk_rms_norm: bad arguments");
    REQUIRE(ctx, N % 256 == 0, "thk_rms_norm: N=%lld must be a multiple of 256 (th.cpp:1155)", (long long)N);
    HIPCHK(ctx, launch_rms_norm(x, (int)rows, (int)N, ctx->stream));
    return THK_OK;
}
extern "C" int thk_row_element_multiply(thk_ctx* ctx, float* x, const float* gain, int64_t rows, int64_t N) {
    if (!ctx) return THK_ERR_INVALID;
    REQUIRE(ctx, x && gain && rows > 0 && N > 0, "thk_row_element_multiply: bad arguments");
    HIPCHK(ctx, launch_row_mul(x, gain, (int)rows, (int)N, ctx->stream));
    return THK_OK;
}
extern "C" int thk_rope(thk_ctx* ctx, float* x, int64_t n_tok, int64_t H, int64_t D, int64_t n_past) {
    if (!ctx) return THK_ERR_INVALID;
    REQUIRE(ctx, x && n_tok > 0 && H > 0 && D > 0 && D % 2 == 0 && n_past >= 0, "thk_rope: bad arguments");
    std::vector<float> tab;
    build_rope_table(tab, (int)D, (int)n_past, (int)n_tok);
    if (ctx->rope_tab_floats < tab.size()) {
        if (ctx->rope_tab) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(ctx->rope_tab)); ctx->rope_tab = nullptr; }
        HIPCHK(ctx, hipMalloc((void**)&ctx->rope_tab, tab.size() * 4));
        ctx->rope_tab_floats = tab.size();
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->rope_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // tab is a stack-lifetime host buffer
    HIPCHK(ctx, launch_rope(x, ctx->rope_tab, (int)n_tok, (int)H, (int)D, 0, ctx->stream));
    return THK_OK;
}
extern "C" int thk_kv_append(thk_ctx* ctx, float* kcache, float* vcache, const float* k, const float* v, int64_t pos, int64_t H, int64_t D) {
    if (!ctx) return THK_ERR_INVALID;
    REQUIRE(ctx, kcache && vcache && k && v && pos >= 0 && H > 0 && D > 0, "thk_kv_append: bad arguments");
    HIPCHK(ctx, launch_kv_append(kcache, vcache, k, v, (int)pos, (int)(H * D), ctx->stream));
    return THK_OK;
}
int valid_head_dim(int64_t D) { return D == 64 || D == 128 || D == 256; }
int valid_splits(int64_t s) { return s == 1 || s == 2 || s == 4 || s == 8; }

extern "C" int thk_attn_decode(thk_ctx* ctx, const float* q, const float* kcache, const float* vcache, int64_t T, int64_t H, int64_t D, float* out) {
    if (!ctx) return THK_ERR_INVALID;
    REQUIRE(ctx, q && kcache && vcache && out && T > 0 && H > 0, "thk_attn_decode: bad arguments");
    REQUIRE(ctx, valid_head_dim(D), "thk_attn_decode: head dim %lld not in {64,128,256}", (long long)D);
    int nsplit = (int)tun(ctx, "attn_splits");
    if (nsplit == 0) nsplit = T > 512 ? 8 : 4;
    REQUIRE(ctx, valid_splits(nsplit), "attn_splits must be 0 (auto), 1, 2, 4 or 8");
    const size_t need = (size_t)H * nsplit * (D + 2) * 4;
    int rc = ensure_scratch(ctx, need < (1u << 20) ? (1u << 20) : need);
    if (rc != THK_OK) return rc;
    AttnArgs a{};
    a.q = q; a.kcache = kcache; a.vcache = vcache; a.pos_ptr = nullptr; a.pos_val = (int)T - 1;
    a.H = (int)H; a.D = (int)D; a.nsplit = nsplit; a.tc = (int)((T + nsplit - 1) / nsplit);
    a.scale = 1.0f / sqrtf((float)D); a.waves = tun(ctx, "attn_waves") == 4 ? 4 : (tun(ctx, "attn_waves") == 16 && D == 128 ? 16 : 8);     // (0 = auto is 8 here: the operator has no cache capacity to go by)
    a.tc_dyn = tun(ctx, "attn_tc_dyn") != 0;
    a.pipe = (T + nsplit - 1) / nsplit > attn_round_positions((int)D, a.waves, false);
    a.part_o = (float*)ctx->scratch; a.part_ml = a.part_o + (size_t)H * nsplit * D;
    a.out = nsplit == 1 ? out : nullptr;
    HIPCHK(ctx, launch_attn_decode(a, ctx->stream));
    if (nsplit > 1) HIPCHK(ctx, launch_attn_combine(a.part_o, a.part_ml, out, (int)H, (int)D, nsplit, ctx->stream));
    return THK_OK;
}
// MFMA tile kernel for D = 64 | 128 (tunable prefill_attn_mfma, default on); otherwise one workgroup per (head, query)
hipError_t attn_prefill_dispatch(thk_ctx* ctx, const float* q, const float* kc, const float* vc, int n_past, int M, int H, int D, float* out, bool kv_f16) {
    if ((D == 64 || D == 128) && tun(ctx, "prefill_attn_mfma") != 0) return launch_attn_prefill_mfma(q, kc, vc, kv_f16, n_past, M, H, D, out, nullptr, ctx->stream);
    AttnArgs a{};
    a.kv_f16 = kv_f16 ? 1 : 0;
    a.q = q; a.kcache = kc; a.vcache = vc; a.pos_ptr = nullptr; a.pos_val = n_past; a.H = H; a.D = D;
    a.nsplit = 1; a.tc = n_past + M; a.scale = 1.0f / sqrtf((float)D); a.waves = 4; a.nq = M; a.out = out;
    return launch_attn_decode(a, ctx->stream);
}
extern "C" int thk_attn_prefill(thk_ctx* ctx, const float* q, const float* kcache, const float* vcache, int64_t n_past, int64_t M, int64_t H, int64_t D, float* out) {
    if (!ctx) return THK_ERR_INVALID;
    REQUIRE(ctx, q && kcache && vcache && out && M > 0 && H > 0 && n_past >= 0, "thk_attn_prefill: bad arguments");
    REQUIRE(ctx, valid_head_dim(D), "thk_attn_prefill: head dim %lld not in {64,128,256}", (long long)D);
    REQUIRE(ctx, n_past + M <= 0x7FFFFFFF / (H * D), "thk_attn_prefill: shape too large");
    HIPCHK(ctx, attn_prefill_dispatch(ctx, q, kcache, vcache, (int)n_past, (int)M, (int)H, (int)D, out));
    return THK_OK;
}
extern "C" int thk_row_softmax(thk_ctx* ctx, float* x, int64_t rows, int64_t N) {
    if (!ctx) return THK_ERR_INVALID;
    REQUIRE(ctx, x && rows > 0 && N > 0, "thk_row_softmax: bad arguments");
    HIPCHK(ctx, launch_row_softmax(x, (int)rows, (int)N, ctx->stream));
    return THK_OK;
}
extern "C" int thk_add(thk_ctx* ctx, const float* a, const float* b, float* c, int64_t n) {
    if (!ctx) return THK_ERR_INVALID;
    REQUIRE(ctx, a && b && c && n > 0, "thk_add: bad arguments");
    HIPCHK(ctx, launch_add(a, b, c, (size_t)n, ctx->stream));
    return THK_OK;
}
extern "C" int thk_silu(thk_ctx* ctx, float* x, int64_t n) {
    if (!ctx) return THK_ERR_INVALID;
    REQUIRE(ctx, x && n > 0, "thk_silu: bad arguments");
    HIPCHK(ctx, launch_silu(x, (size_t)n, ctx->stream));
    return THK_OK;
}
extern "C" int thk_mul_inplace(thk_ctx* ctx, float* a, const float* b, int64_t n) {
    if (!ctx) return THK_ERR_INVALID;
    REQUIRE(ctx, a && b && n > 0, "thk_mul_inplace: bad arguments");
    HIPCHK(ctx, launch_mul(a, b, (size_t)n, ctx->stream));
    return THK_OK;
}
void q1_constants(int V, int* split, int* cov) {   // th.cpp:3990-3996 with numSplits = 8 (th-llama.cpp:262)
    int s = V / 8; if (s < 1) s = 1;
    int kTile = s / 256; if (kTile == 0) kTile = 1;
    int c = 256 * kTile; if (c > s) c = s;
    *split = s; *cov = c;
}
extern "C" int thk_lmhead_f16(thk_ctx* ctx, const void* W, int64_t V, int64_t E, const float* x, float* logits, int mode) {
    if (!ctx) return THK_ERR_INVALID;
    REQUIRE(ctx, W && x && logits && V > 0, "thk_lmhead_f16: bad arguments");
    REQUIRE(ctx, E >= 512 && E % 512 == 0, "thk_lmhead_f16: E=%lld must be a multiple of 512 (th.cpp:3728-3739)", (long long)E);
    REQUIRE(ctx, mode == THK_LMHEAD_CORRECT || mode == THK_LMHEAD_FAITHFUL, "thk_lmhead_f16: bad mode");
    int rc = ensure_scratch(ctx, 1u << 20);
    if (rc != THK_OK) return rc;
    GemvArgs a{}; a.W[0] = (const uint16_t*)W; a.R = (int)V; a.C = (int)E; a.x = x; a.y = logits;
    a.lm_faithful = mode == THK_LMHEAD_FAITHFUL; q1_constants((int)V, &a.q1_split, &a.q1_cov);
    a.block_best = (unsigned long long*)ctx->scratch;
    rc = gemv_simple(ctx, GEMV_PRO_COPY, GEMV_EPI_HEAD, "gemv_variant_head", "gemv_bpc_head", a, (int)V);
    return rc < 0 ? rc : THK_OK;
}
extern "C" int thk_argmax(thk_ctx* ctx, const float* logits, int64_t V, int32_t* id_out) {
    if (!ctx) return THK_ERR_INVALID;
    REQUIRE(ctx, logits && id_out && V > 0, "thk_argmax: bad arguments");
    int rc = ensure_scratch(ctx, 1u << 20);
    if (rc != THK_OK) return rc;
    int nblocks = (int)((V + kBlock - 1) / kBlock); if (nblocks > 256) nblocks = 256;
    HIPCHK(ctx, launch_argmax(logits, (int)V, (unsigned long long*)ctx->scratch, nblocks, ctx->stream));
    HIPCHK(ctx, launch_finish_token((const unsigned long long*)ctx->scratch, nblocks, nullptr, nullptr, 0, nullptr, id_out, 0, nullptr, ctx->stream));
    return THK_OK;
}
// The k largest entries of a device logits vector, value descending, ties by ascending index (a total order).  Selection and sort
// run on the GPU (launch_topk); k x 8 bytes come back instead of V x 4.  Feeds the host sampler's top-k stage
// (th-llama.cpp:814-907 partial-sorts all n_vocab candidates on the CPU after a 128 KB read-back).
int topk_to_host(thk_ctx* ctx, const float* logits_dev, int64_t V, int32_t k, float* values_out, int32_t* ids_out) {
    REQUIRE(ctx, logits_dev && values_out && ids_out, "top-k: null argument");
    REQUIRE(ctx, V >= 1 && V <= 32768 && k >= 1 && k <= 1024 && k <= V, "top-k: V=%lld k=%d outside the device kernel's range (V <= 32768, k <= 1024)", (long long)V, k);
    int rc = ensure_scratch(ctx, 1u << 20);
    if (rc != THK_OK) return rc;
    unsigned long long* keys_dev = (unsigned long long*)ctx->scratch;             // [0, 8 KiB): the k keys; behind them the local candidates (<= 256 KiB)
    HIPCHK(ctx, launch_topk(logits_dev, (int)V, k, keys_dev, keys_dev + 1024, ctx->stream));
    std::vector<unsigned long long> keys((size_t)k);
    HIPCHK(ctx, hipMemcpyAsync(keys.data(), keys_dev, (size_t)k * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    topk_decode_keys(keys.data(), k, values_out, ids_out);
    return THK_OK;
}
void topk_decode_keys(const unsigned long long* keys, int32_t k, float* values_out, int32_t* ids_out) {
    for (int i = 0; i < k; ++i) {
        const unsigned hi = (unsigned)(keys[i] >> 32), lo = (unsigned)(keys[i] & 0xFFFFFFFFull);
        const unsigned bits = (hi & 0x80000000u) ? (hi & 0x7FFFFFFFu) : ~hi;     // inverse of argmax_key's order-preserving map
        memcpy(&values_out[i], &bits, 4);
        ids_out[i] = (int32_t)(0xFFFFFFFFu - lo);
    }
}
// The top-k kernel enqueued behind whatever the stream holds, writing its keys straight into a host-mapped page: the caller
// synchronises the stream ONCE (for the step and the selection together) and decodes ctx->pinned_keys.
int topk_enqueue_pinned(thk_ctx* ctx, const float* logits_dev, int64_t V, int32_t k) {
    REQUIRE(ctx, logits_dev && V >= 1 && V <= 32768 && k >= 1 && k <= 1024 && k <= V, "top-k: V=%lld k=%d outside the device kernel's range (V <= 32768, k <= 1024)", (long long)V, k);
    if (!ctx->pinned_keys) {
        HIPCHK(ctx, hipHostMalloc((void**)&ctx->pinned_keys, 1025 * 8, hipHostMallocMapped));
        HIPCHK(ctx, hipHostGetDevicePointer((void**)&ctx->pinned_keys_dev, ctx->pinned_keys, 0));
        ctx->pinned_keys[1024] = 0;
    }
    const int rc = ensure_scratch(ctx, 1u << 20);
    if (rc != THK_OK) return rc;
    HIPCHK(ctx, launch_topk(logits_dev, (int)V, k, ctx->pinned_keys_dev, (unsigned long long*)ctx->scratch + 1024, ctx->stream, ctx->pinned_keys_dev + 1024, ++ctx->topk_epoch));
    return THK_OK;
}
// The kernel's last act is a system-scope store of this call's epoch behind its keys: the host thread polls that word in its own memory
// (a hipStreamSynchronize wake-up costs tens of microseconds per token); after 0.2 s of polling it falls back to the stream, so a fault
// still surfaces as an error.  ctx->pinned_keys / topk_epoch are per CONTEXT and unlocked: like every other entry point (thk.h: "thread-compatible, not
// thread-safe") thk_model_eval_topk is for one thread per context at a time - two models on one context take turns.
int topk_wait_pinned(thk_ctx* ctx) {
    volatile unsigned long long* stamp = ctx->pinned_keys + 1024;
    const unsigned long long want = ctx->topk_epoch;
    timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 0; *stamp != want; ++spins) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        sched_yield();
#endif
        if ((spins & 0xFFFu) == 0xFFFu) {
            timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9 > 0.2) {
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                REQUIRE(ctx, *stamp == want, "top-k: the kernel finished without publishing its stamp");
                break;
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return THK_OK;
}
extern "C" int thk_topk_f32(thk_ctx* ctx, const float* logits, int64_t V, int32_t k, float* values_out, int32_t* ids_out) {
    if (!ctx) return THK_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return topk_to_host(ctx, logits, V, k, values_out, ids_out);
}
extern "C" int thk_embed_f16(thk_ctx* ctx, const void* table, int64_t E, int32_t token, float* x) {
    if (!ctx) return THK_ERR_INVALID;
    REQUIRE(ctx, table && x && E > 0 && token >= 0, "thk_embed_f16: bad arguments");
    HIPCHK(ctx, launch_embed((const uint16_t*)table, nullptr, token, (int)E, x, ctx->stream));
    return THK_OK;
}
extern "C" int thk_synth_f16(thk_ctx* ctx, const char* name, uint64_t seed, float sigma, int64_t n, void* out) {
    if (!ctx) return THK_ERR_INVALID;
    REQUIRE(ctx, name && out && n > 0, "thk_synth_f16: bad arguments");
    HIPCHK(ctx, launch_synth_f16(synth_key(name, seed), synth_scale(sigma), (size_t)n, out, ctx->stream));
    return THK_OK;
}
extern "C" int thk_synth_gain_f32(thk_ctx* ctx, const char* name, uint64_t seed, float sigma, int64_t n, float* out) {
    if (!ctx) return THK_ERR_INVALID;
    REQUIRE(ctx, name && out && n > 0, "thk_synth_gain_f32: bad arguments");
    HIPCHK(ctx, launch_synth_gain(synth_key(name, seed), synth_scale(sigma), (size_t)n, out, ctx->stream));
    return THK_OK;
}
extern "C" int thk_gemm_f16_prefill(thk_ctx* ctx, const void* W, int64_t R, int64_t C, const float* X, int64_t M, float* Y) {
    if (!ctx) return THK_ERR_INVALID;
    REQUIRE(ctx, W && X && Y && R > 0 && M > 0, "thk_gemm_f16_prefill: bad arguments");
    REQUIRE(ctx, C >= 32 && C % 32 == 0, "thk_gemm_f16_prefill: C=%lld must be a multiple of 32", (long long)C);
    const size_t ws = gemm_prefill_workspace_bytes((int)M, (int)R, (int)C);
    int rc = ensure_scratch(ctx, ws < (1u << 20) ? (1u << 20) : ws);
    if (rc != THK_OK) return rc;
    HIPCHK(ctx, launch_gemm_f16_prefill((const uint16_t*)W, (int)R, (int)C, X, (int)M, Y, ctx->scratch, ctx->stream));
    return THK_OK;
}
