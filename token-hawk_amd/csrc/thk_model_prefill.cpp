// thk_model_prefill.cpp — thk_model_prefill: the batched prompt pass (config C3) on the MFMA GEMM path of thk_prefill.hip.
#include "thk_internal.hpp"

// Batched prompt prefill (config C3): the M prompt tokens go through the layers together so every
// weight matrix is streamed ONCE and multiplied on the matrix cores (thk_prefill.hip), instead of
// M mat-vec passes.  Semantics == feeding the tokens one at a time (the reference's own batch path
// is disabled, th-llama.cpp:15, and its causal mask is only right at n_past == 0, Q5): causal
// attention over the f32 cache, K/V rows appended at [n_past, n_past+M), logits of the last token.
//
// Per 128-token slab and layer, 9 launches (round 4; 11 before): wq,wk,wv GEMM | reduce + 1/rms + RoPE + KV write | causal
// attention -> X image | wo GEMM | reduce + residual -> x, image of x * ffn gain, sum of squares | w1,w3 GEMM | reduce + 1/rms +
// SwiGLU -> X image | w2 GEMM | reduce + residual -> x, image of x * the NEXT layer's attention gain, sum of squares.  The two
// norm -> image launches are gone: RMSNorm's per-token scalar is applied on the output side of the GEMM ("deferred norm",
// thk_prefill.hip; tunable prefill_deferred_norm = 0 restores the 11-launch form).  Layer 0's image comes from the embedding rows.
// (thk_prefill.hip explains the GEMM: LDS-DMA pipeline, stream-K, hi/lo split.)
// The four GEMM plans of a slab of M tokens (qkv, wo, w13, w2) from the prefill_blocks_* / prefill_tile_* tunables.
static int slab_plans(thk_model* m, int M, PrefillPlan out[4]) {
    thk_ctx* ctx = m->ctx;
    const int E = m->hp.n_embd, F = m->n_ff;
    static const char* const kind[4] = {"qkv", "wo", "w13", "w2"};
    const int R[4] = {E, E, F, E}, nmat[4] = {3, 1, 2, 1}, C[4] = {E, E, E, F};
    for (int k = 0; k < 4; ++k) {
        int g = (int)tun(ctx, (std::string("prefill_blocks_") + kind[k]).c_str());
        const int t = (int)tun(ctx, (std::string("prefill_tile_") + kind[k]).c_str());
        REQUIRE(ctx, g >= 0 && g <= 256, "prefill_blocks_* tunables must be in [0 (auto), 256]");
        if (g == 0) {
            // auto (round 4): stream-K shares that do not straddle row-blocks - w workgroups per row-block with w | chunks per
            // row-block - spill one partial tile per workgroup instead of 1.3, but only if that keeps >= 192 of the 256 CUs busy
            // (7B wq|wk|wv: 48 row-blocks x 4 = 192 workgroups: -1.5 % prefill time; w1|w3: 86 x 2 = 172 measured slower than 256)
            const PrefillPlan probe = prefill_plan(M, R[k], nmat[k], C[k], 256, t);
            g = 256;
            for (int w = 256 / probe.rb_total; w >= 1; --w)
                if (probe.nchunks % w == 0 && probe.rb_total * w >= 192) { g = probe.rb_total * w; break; }
        }
        out[k] = prefill_plan(M, R[k], nmat[k], C[k], g, t);
    }
    return THK_OK;
}
struct PrefillBufs { float *X, *Q, *ATT; int32_t* tok; char *imgE, *imgF; float* part; unsigned long long* ssq; size_t ssq_bytes; };
static size_t align256(size_t v) { return (v + 255) / 256 * 256; }
// tokens per slab = tokens per weight pass: 256 (round 5: gemm_prefill_v3h_kernel, eight token tiles) or 128 (tunable prefill_slab_tokens)
static int slab_tokens(thk_model* m) { return tun(m->ctx, "prefill_slab_tokens") == 128 ? 128 : 256; }
static int prefill_workspace(thk_model* m, PrefillBufs* b) {
    thk_ctx* ctx = m->ctx;
    const int E = m->hp.n_embd;
    const int SL = slab_tokens(m);
    const size_t per = align256((size_t)SL * E * 4);
    // sized from the SAME plans prefill_slab builds (the prefill_blocks_* / prefill_tile_* tunables are read per call, and
    // part_floats = G * maxseg * slot_floats is not monotonic in G, so a fixed G = 256 bound could be exceeded)
    size_t part_floats = 0, img_e = 0, img_f = 0;
    for (int M : {128, SL}) {
        PrefillPlan pl[4];
        const int rc = slab_plans(m, M, pl);
        if (rc != THK_OK) return rc;
        for (int k = 0; k < 4; ++k) part_floats = std::max(part_floats, pl[k].part_floats);
        img_e = std::max(img_e, std::max(std::max(pl[0].ximg_bytes, pl[1].ximg_bytes), pl[2].ximg_bytes));
        img_f = std::max(img_f, pl[3].ximg_bytes);
    }
    const size_t imgE = align256(img_e), imgF = align256(img_f), part = align256(4 * part_floats);
    const size_t ssq_bytes = align256((size_t)(m->l1 - m->l0) * 2 * 256 * 8);      // deferred norm: sum of squares per (layer, norm, token of the slab), 2^-32 fixed point
    const size_t bytes = 3 * per + 1024 + imgE + imgF + part + ssq_bytes;
    if (m->prefill_ws_bytes < bytes) {
        if (m->prefill_ws) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(m->prefill_ws)); m->prefill_ws = nullptr; }
        hipError_t e = hipMalloc(&m->prefill_ws, bytes);
        if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? THK_ERR_OOM : THK_ERR_HIP, "prefill workspace (%zu bytes): %s", bytes, hipGetErrorString(e));
        m->prefill_ws_bytes = bytes;
    }
    char* p = (char*)m->prefill_ws;
    b->X = (float*)p; p += per; b->Q = (float*)p; p += per; b->ATT = (float*)p; p += per;
    b->tok = (int32_t*)p; p += 1024; b->imgE = p; p += imgE; b->imgF = p; p += imgF; b->part = (float*)p; p += part;
    b->ssq = (unsigned long long*)p; b->ssq_bytes = ssq_bytes;
    return THK_OK;
}

// Tile images of this stage's layer matrices (thk_prefill.hip, pack_w_kernel): a second copy of the layer weights in HBM (12.4 GB
// for 7B of 288), made by thk_model_prepare_prefill - or lazily by the first thk_model_prefill - and again when a weight or a
// prefill_tile_* tunable changed.  If the slab does not fit, the GEMMs stay on the row-major matrices (still the HIP path, ~20 %
// slower): thk_model_prefill_uses_tile_images reports which, thk_last_error says why, and the next weight write, tunable change or
// explicit prepare call tries again.
static int ensure_prefill_pack(thk_model* m, bool explicit_call) {
    thk_ctx* ctx = m->ctx;
    if (tun(ctx, "prefill_packed") == 0) return THK_OK;
    if (m->pk_failed && !explicit_call) return THK_OK;
    const int E = m->hp.n_embd, F = m->n_ff, nl = m->l1 - m->l0;
    int tiles[4];
    {
        PrefillPlan pl[4];
        const int rc = slab_plans(m, 128, pl);           // the tiles a full slab uses
        if (rc != THK_OK) return rc;
        for (int k = 0; k < 4; ++k) tiles[k] = pl[k].tile_rows;
    }
    if (m->prefill_pk && !m->pk_w.empty() && !memcmp(tiles, m->pk_tiles, sizeof tiles)) return THK_OK;
    // wq wk wv wo w1 w2 w3: (rows, cols, tile)
    const int R[7] = {E, E, E, E, F, E, F}, C[7] = {E, E, E, E, E, F, E}, T[7] = {tiles[0], tiles[0], tiles[0], tiles[1], tiles[2], tiles[3], tiles[2]};
    size_t per_layer = 0, off[7];
    const size_t kAlign = (size_t)2 << 20;                  // 2 MiB, as the weight slab (a 1 MiB phase costs the decode kernels 2 %)
    for (int k = 0; k < 7; ++k) { off[k] = per_layer; per_layer += (prefill_pack_bytes(R[k], C[k], T[k]) + kAlign - 1) / kAlign * kAlign; }
    const size_t bytes = per_layer * nl;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (m->prefill_pk_bytes < bytes) {
        hipFree(m->prefill_pk); m->prefill_pk = nullptr; m->prefill_pk_bytes = 0;
        if (hipMalloc(&m->prefill_pk, bytes) != hipSuccess) {
            (void)hipGetLastError(); m->prefill_pk = nullptr; m->pk_failed = true; m->pk_w.clear(); m->pk_tiles[0] = 0;
            (void)fail(ctx, THK_ERR_OOM, "prefill tile images (%zu bytes) do not fit: the prefill GEMMs stream the row-major matrices instead (~20 %% slower)", bytes);
            return explicit_call ? THK_ERR_OOM : THK_OK;      // the lazy path degrades (query: thk_model_prefill_uses_tile_images), an explicit prepare reports
        }
        m->prefill_pk_bytes = bytes;
    }
    m->pk_w.assign(nl, {});
    for (int i = 0; i < nl; ++i) {
        const LayerW& L = m->layers[i];
        const uint16_t* src[7] = {L.wq, L.wk, L.wv, L.wo, L.w1, L.w2, L.w3};
        for (int k = 0; k < 7; ++k) {
            char* dst = (char*)m->prefill_pk + (size_t)i * per_layer + off[k];
            HIPCHK(ctx, launch_prefill_pack(src[k], R[k], C[k], T[k], dst, ctx->stream));
            m->pk_w[i][k] = reinterpret_cast<const uint16_t*>(dst);
        }
    }
    memcpy(m->pk_tiles, tiles, sizeof tiles);
    m->pk_failed = false;
    return THK_OK;
}
// Pay for the prefill path now (workspace + tile images: time and HBM) instead of inside the first thk_model_prefill call.
extern "C" int thk_model_prepare_prefill(thk_model* m) {
    if (!m) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    REQUIRE(ctx, m->finalized, "thk_model_prepare_prefill before thk_model_finalize");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PrefillBufs b{};
    int rc = prefill_workspace(m, &b);
    if (rc != THK_OK) return rc;
    rc = ensure_prefill_pack(m, true);
    if (rc == THK_OK) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return rc;
}
// 1: the next thk_model_prefill streams tile images; 0: row-major matrices (not prepared yet, prefill_packed = 0, or the slab did not fit)
extern "C" int thk_model_prefill_uses_tile_images(const thk_model* m) { return (m && m->prefill_pk && !m->pk_w.empty() && !m->pk_failed) ? 1 : 0; }

// one slab of M <= 256 prompt tokens at positions [n_past, n_past + M) through every layer
// hidden (device f32 [M, E], may be null on a full-model stage): the slab's rows of the stage hand-off buffer - read as the stage input when this
// stage has no embedding table, written with the stage output when it has no head (in place: a stage turns its input rows into its output rows)
static int prefill_slab(thk_model* m, SeqBuf& sb, const PrefillBufs& b, const int32_t* tokens, float* hidden, int M, int n_past) {
    thk_ctx* ctx = m->ctx;
    hipStream_t st = ctx->stream;
    const int E = m->hp.n_embd, H = m->hp.n_head, D = E / H, F = m->n_ff, T = m->hp.n_ctx;
    PrefillPlan pl[4];
    {
        const int rc = slab_plans(m, M, pl);
        if (rc != THK_OK) return rc;
    }
    PrefillPlan &pq = pl[0], &po = pl[1], &p13 = pl[2], &p2 = pl[3];
    const bool pk = !m->pk_w.empty() && m->pk_tiles[0] == pq.tile_rows && m->pk_tiles[1] == po.tile_rows && m->pk_tiles[2] == p13.tile_rows && m->pk_tiles[3] == p2.tile_rows;
    pq.packed = po.packed = p13.packed = p2.packed = (pk ? 1 : 0) | (tun(ctx, "prefill_wave_grid") != 0 ? 2 : 0);     // bit 1: 256-token slabs on the 2 x 2 wave grid (gemm_prefill_v3g_kernel)
    if (m->flags & THK_STAGE_EMBED) {
        HIPCHK(ctx, hipMemcpyAsync(b.tok, tokens, (size_t)M * 4, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipStreamSynchronize(st));   // tokens may be a stack buffer
        HIPCHK(ctx, launch_embed_rows(m->tok_embeddings, b.tok, M, E, b.X, st));
    } else {
        HIPCHK(ctx, hipMemcpyAsync(b.X, hidden, (size_t)M * E * 4, hipMemcpyDeviceToDevice, st));     // the previous stage's output rows (th-llama.cpp:305-311 is per layer: nothing ties the batch branch to a whole model)
    }
    const int nl = m->l1 - m->l0;
    const bool defer = tun(ctx, "prefill_deferred_norm") != 0;
    if (defer) HIPCHK(ctx, hipMemsetAsync(b.ssq, 0, b.ssq_bytes, st));
    auto ssq_of = [&](int layer, int which) { return b.ssq + ((size_t)layer * 2 + which) * 256; };      // which: 0 attention norm, 1 ffn norm
    for (int i = 0; i < nl; ++i) {
        const LayerW& L = m->layers[i];
        float* kc = kcache_of(m, sb, i);
        float* vc = vcache_of(m, sb, i);
        const uint16_t* wqkv[3] = {pk ? m->pk_w[i][0] : L.wq, pk ? m->pk_w[i][1] : L.wk, pk ? m->pk_w[i][2] : L.wv};
        const uint16_t* w13[2] = {pk ? m->pk_w[i][4] : L.w1, pk ? m->pk_w[i][6] : L.w3};
        const uint16_t* wo = pk ? m->pk_w[i][3] : L.wo;
        const uint16_t* w2 = pk ? m->pk_w[i][5] : L.w2;
        if (!defer) HIPCHK(ctx, launch_prefill_ximg(b.X, L.attention_norm, M, E, b.imgE, st));
        else if (i == 0) HIPCHK(ctx, launch_prefill_ximg(b.X, L.attention_norm, M, E, b.imgE, st, ssq_of(0, 0)));     // later layers: written by the previous layer's last reducer
        HIPCHK(ctx, launch_prefill_gemm(wqkv, pq, b.imgE, b.part, st));
        // the image's power-of-two scale came from: the row itself (layer 0), else the previous layer's ffn-norm input
        HIPCHK(ctx, launch_prefill_reduce_qkv(b.part, pq, m->rope_tab, n_past, D, b.Q, kc, vc, m->kv_f16 != 0, st, defer ? ssq_of(i, 0) : nullptr,
                                              defer ? (i == 0 ? ssq_of(0, 0) : ssq_of(i - 1, 1)) : nullptr));
        if ((D == 64 || D == 128) && tun(ctx, "prefill_attn_mfma") != 0) {
            if (M <= 128) {
                HIPCHK(ctx, launch_attn_prefill_mfma(b.Q, kc, vc, m->kv_f16 != 0, n_past, M, H, D, nullptr, b.imgE, st));   // writes wo's X image directly
            } else {      // a 256-token slab: one launch, eight 32-query tiles per head (256 workgroups for 7B), the slab's eight-tile image (pad tiles as zero rows)
                HIPCHK(ctx, launch_attn_prefill_mfma(b.Q, kc, vc, m->kv_f16 != 0, n_past, M, H, D, nullptr, b.imgE, st, 8, 0, 8));
            }
        } else {
            HIPCHK(ctx, attn_prefill_dispatch(ctx, b.Q, kc, vc, n_past, M, H, D, b.ATT, m->kv_f16 != 0));
            HIPCHK(ctx, launch_prefill_ximg(b.ATT, nullptr, M, E, b.imgE, st));
        }
        HIPCHK(ctx, launch_prefill_gemm(&wo, po, b.imgE, b.part, st));
        if (defer) {
            HIPCHK(ctx, launch_prefill_reduce_resid_ximg(b.part, po, b.X, L.ffn_norm, b.imgE, ssq_of(i, 1), ssq_of(i, 0), st));
        } else {
            HIPCHK(ctx, launch_prefill_reduce_store(b.part, po, b.X, true, st));
            HIPCHK(ctx, launch_prefill_ximg(b.X, L.ffn_norm, M, E, b.imgE, st));
        }
        HIPCHK(ctx, launch_prefill_gemm(w13, p13, b.imgE, b.part, st));
        HIPCHK(ctx, launch_prefill_reduce_swiglu(b.part, p13, b.imgF, st, defer ? ssq_of(i, 1) : nullptr, defer ? ssq_of(i, 0) : nullptr));
        HIPCHK(ctx, launch_prefill_gemm(&w2, p2, b.imgF, b.part, st));
        if (defer && i + 1 < nl) HIPCHK(ctx, launch_prefill_reduce_resid_ximg(b.part, p2, b.X, m->layers[i + 1].attention_norm, b.imgE, ssq_of(i + 1, 0), ssq_of(i, 1), st));
        else HIPCHK(ctx, launch_prefill_reduce_store(b.part, p2, b.X, true, st));
    }
    if (hidden && !(m->flags & THK_STAGE_HEAD)) HIPCHK(ctx, hipMemcpyAsync(hidden, b.X, (size_t)M * E * 4, hipMemcpyDeviceToDevice, st));
    return THK_OK;
}

// The prompt pass of ONE pipeline stage (config C3 x C4): this stage's layers [layer_begin, layer_end) over the n_tokens prompt rows, slab by slab.
// The batch branch of the reference is per layer (th-llama.cpp:305-311, :365-404), so a contiguous layer range is as good a unit as the whole model.
static int prefill_stage(thk_model* m, int32_t seq, const int32_t* tokens, float* hidden_dev, int32_t n_tokens, int32_t n_past, float* logits_out, const char* who) {
    if (!m) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    const bool embed = m->flags & THK_STAGE_EMBED, head = m->flags & THK_STAGE_HEAD;
    REQUIRE(ctx, m->finalized, "%s before thk_model_finalize", who);
    REQUIRE(ctx, seq >= 0 && seq < m->n_seq, "bad sequence %d", seq);
    REQUIRE(ctx, !embed || tokens, "%s: an embedding stage needs token ids", who);
    REQUIRE(ctx, (embed && head) || hidden_dev, "%s: a stage without the embedding table reads its input rows from hidden_dev, a stage without the head leaves its output rows there", who);
    REQUIRE(ctx, !logits_out || head, "%s: logits requested from a stage without the lm-head", who);
    REQUIRE(ctx, n_tokens >= 1 && n_past >= 0 && n_past + n_tokens <= m->hp.n_ctx, "n_past=%d + n_tokens=%d exceeds n_ctx=%d", n_past, n_tokens, m->hp.n_ctx);
    REQUIRE(ctx, m->hp.n_embd % 32 == 0 && m->n_ff % 32 == 0, "%s needs n_embd and n_ff to be multiples of 32", who);
    if (embed) for (int i = 0; i < n_tokens; ++i) REQUIRE(ctx, tokens[i] >= 0 && tokens[i] < m->hp.n_vocab, "token id %d out of range", tokens[i]);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PrefillBufs b{};
    int rc = prefill_workspace(m, &b);
    if (rc != THK_OK) return rc;
    if ((rc = ensure_prefill_pack(m, false)) != THK_OK) return rc;
    hipStream_t st = ctx->stream;
    SeqBuf& sb = m->seqs[seq];
    const int E = m->hp.n_embd, V = m->hp.n_vocab, M = n_tokens;
    int last = 0;
    const int SL = slab_tokens(m);
    for (int m0 = 0; m0 < M; m0 += last) {    // slabs of <= 256 tokens (one weight pass each); later slabs attend to the rows earlier ones cached
        const int left = M - m0;
        last = left > 128 ? std::min(SL, left) : left;       // 129 .. 256 tokens left: ONE eight-tile pass (pad tiles are cheaper than a second weight pass)
        if ((rc = prefill_slab(m, sb, b, embed ? tokens + m0 : nullptr, hidden_dev ? hidden_dev + (size_t)m0 * E : nullptr, last, n_past + m0)) != THK_OK) return rc;
    }
    if (head) {   // final norm + lm-head on the last token only (th-llama.cpp:253-262, aOffset = (r-1)*c)
        GemvArgs a{};
        a.W[0] = m->output; a.R = V; a.C = E;
        const int NR = gemv_rows_per_group(E, GEMV_EPI_HEAD, m->var_head);
        a.n_groups = (V + NR - 1) / NR;
        a.x = b.X + (size_t)(last - 1) * E; a.gain = m->norm; a.y = sb.logits;
        a.lm_faithful = m->lm_mode == THK_LMHEAD_FAITHFUL; q1_constants(V, &a.q1_split, &a.q1_cov);
        a.block_best = m->block_best_aux;      // not the decode step's slots: those stay zero between launches (folded greedy pick)
        HIPCHK(ctx, launch_gemv(GEMV_PRO_RMS, GEMV_EPI_HEAD, m->var_head, a, m->grid_head, m->nt != 0, st));
        HIPCHK(ctx, hipMemcpyAsync(m->x, b.X + (size_t)(last - 1) * E, (size_t)E * 4, hipMemcpyDeviceToDevice, st));
    }
    rc = set_seq_state(m, seq, embed ? tokens[M - 1] : 0, n_past + M - 1, false);
    if (rc != THK_OK) return rc;
    if (logits_out) HIPCHK(ctx, hipMemcpyAsync(logits_out, sb.logits, (size_t)V * 4, hipMemcpyDeviceToHost, st));
    if (logits_out || (embed && head)) HIPCHK(ctx, hipStreamSynchronize(st));     // a stage call without read-back stays stream-ordered: the hand-off behind it is too
    return THK_OK;
}

extern "C" int thk_model_prefill(thk_model* m, int32_t seq, const int32_t* tokens, int32_t n_tokens, int32_t n_past, float* logits_out) {
    if (!m) return THK_ERR_INVALID;
    REQUIRE(m->ctx, (m->flags & THK_STAGE_EMBED) && (m->flags & THK_STAGE_HEAD), "thk_model_prefill needs a full-model stage (embedding + head); pipeline stages call thk_model_prefill_stage");
    REQUIRE(m->ctx, tokens, "bad sequence %d / null tokens", seq);
    return prefill_stage(m, seq, tokens, nullptr, n_tokens, n_past, logits_out, "thk_model_prefill");
}
extern "C" int thk_model_prefill_stage(thk_model* m, int32_t seq, const int32_t* tokens, float* hidden_dev, int32_t n_tokens, int32_t n_past, float* logits_out) {
    return prefill_stage(m, seq, tokens, hidden_dev, n_tokens, n_past, logits_out, "thk_model_prefill_stage");
}
