// thk_model.cpp — the model level of the C-ABI: model objects (weight slab, working buffers, per-sequence state), tensor upload,
// finalize (launch geometry, buffers, the per-sequence graphs), th_eval_gpu's entry point thk_model_eval (th-llama.cpp:464-660) and
// the sequence accessors.  The decode step itself - what is launched, the hipGraph replays, the step-level API - lives in
// thk_model_step.cpp, the optional one-launch engine's program builder in thk_model_engine.cpp, the MFMA prompt prefill in
// thk_model_prefill.cpp.
#include "thk_internal.hpp"

// ---------------------------------------------------------------- model
static int n_ff_of(const thk_hparams& hp) { return ((2 * (4 * hp.n_embd) / 3 + hp.n_mult - 1) / hp.n_mult) * hp.n_mult; }   // loader :349

extern "C" int thk_model_create(thk_ctx* ctx, const thk_hparams* hp, int32_t layer_begin, int32_t layer_end, uint32_t stage_flags,
                                int32_t n_seq, thk_model** out) {
    if (!ctx || !hp || !out) return THK_ERR_INVALID;
    *out = nullptr;
    REQUIRE(ctx, hp->n_embd >= 512 && hp->n_embd % 512 == 0, "n_embd=%d must be a multiple of 512 (th.cpp:3728-3739)", hp->n_embd);
    REQUIRE(ctx, hp->n_head > 0 && hp->n_embd % hp->n_head == 0, "n_head must divide n_embd");
    REQUIRE(ctx, valid_head_dim(hp->n_embd / hp->n_head), "head dim %d not in {64,128,256}", hp->n_embd / hp->n_head);
    REQUIRE(ctx, hp->n_mult > 0 && n_ff_of(*hp) % 256 == 0, "n_ff=%d must be a multiple of 256", n_ff_of(*hp));
    REQUIRE(ctx, hp->n_layer > 0 && layer_begin >= 0 && layer_begin < layer_end && layer_end <= hp->n_layer, "bad layer range [%d,%d)", layer_begin, layer_end);
    REQUIRE(ctx, hp->n_vocab > 0 && hp->n_ctx > 0 && n_seq >= 1 && n_seq <= 64, "bad n_vocab / n_ctx / n_seq");
    REQUIRE(ctx, !(stage_flags & THK_STAGE_EMBED) || layer_begin == 0, "the embedding stage must start at layer 0");
    REQUIRE(ctx, !(stage_flags & THK_STAGE_HEAD) || layer_end == hp->n_layer, "the head stage must end at the last layer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    thk_model* m = new thk_model();
    m->ctx = ctx; m->hp = *hp; m->n_ff = n_ff_of(*hp); m->l0 = layer_begin; m->l1 = layer_end; m->flags = stage_flags; m->n_seq = n_seq;
    const size_t E = hp->n_embd, F = m->n_ff, V = hp->n_vocab;
    m->layers.resize(layer_end - layer_begin);
    // One slab for every weight of the stage, tensors in the order a decode step reads them, each on a 2 MiB boundary: one
    // mapping with the largest page fragments the driver can give (a decode step walks 13.5 GB through ~160 kernels, each of
    // which starts on a matrix it has not touched since the previous token), instead of ~290 separate allocations:
    // +1 % decode tokens/s (395.9 -> 399.3-400.6, profiles/r02_weight_slab.txt).
    const size_t kAlign = (size_t)2 << 20;     // measured on MI355X: 2-16 MiB equal, 1 MiB -2 %, 512 MiB -1 % (every matrix then starts on the same channels)
    auto up = [&](size_t b) { return (b + kAlign - 1) / kAlign * kAlign; };
    size_t total = 0;
    auto reserve = [&](size_t bytes) { const size_t off = total; total += up(bytes); return off; };
    struct Slot { void** p; size_t off; };
    std::vector<Slot> slots;
#define ALLOC(ptr, bytes) slots.push_back(Slot{(void**)&(ptr), reserve(bytes)})
    if (stage_flags & THK_STAGE_EMBED) ALLOC(m->tok_embeddings, V * E * 2);
    for (auto& L : m->layers) {
        ALLOC(L.attention_norm, E * 4);
        ALLOC(L.wq, E * E * 2); ALLOC(L.wk, E * E * 2); ALLOC(L.wv, E * E * 2); ALLOC(L.wo, E * E * 2);
        ALLOC(L.ffn_norm, E * 4);
        ALLOC(L.w1, F * E * 2); ALLOC(L.w3, F * E * 2); ALLOC(L.w2, E * F * 2);
    }
    if (stage_flags & THK_STAGE_HEAD) { ALLOC(m->norm, E * 4); ALLOC(m->output, V * E * 2); }
#undef ALLOC
    {
        hipError_t e_ = hipMalloc(&m->weights_slab, total);
        if (e_ != hipSuccess) { int rc_ = fail(ctx, e_ == hipErrorOutOfMemory ? THK_ERR_OOM : THK_ERR_HIP, "hipMalloc(%zu) for the stage's weights: %s", total, hipGetErrorString(e_)); thk_model_destroy(m); return rc_; }
        m->weights_slab_bytes = total;
        for (auto& sl : slots) *sl.p = (char*)m->weights_slab + sl.off;
    }
    *out = m;
    return THK_OK;
}

// Working buffers (activation vectors, attention partials, RoPE table, per-sequence state, logits ...) are carved from 8 MiB
// chunks instead of ~25 small hipMallocs: every decode kernel's prologue starts with a dependent read of two or three of them,
// and buffers this small would otherwise sit on pages of their own (measured: neutral for tokens/s, 399.5 vs 399.2 — kept for
// the single release point and the faster finalize).  The KV cache, the weights and the prefill buffers keep
// their own large allocations.  Everything carved here is released by free_working() in one go.
static int arena_alloc(thk_model* m, void** out, size_t bytes) {
    thk_ctx* ctx = m->ctx;
    const size_t need = (bytes + 255) / 256 * 256;
    if (m->arena_chunks.empty() || m->arena_off + need > m->arena_cap) {
        const size_t cap = std::max((size_t)8 << 20, (need + ((size_t)2 << 20) - 1) / ((size_t)2 << 20) * ((size_t)2 << 20));
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, cap);
        if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? THK_ERR_OOM : THK_ERR_HIP, "hipMalloc(%zu) for working buffers: %s", cap, hipGetErrorString(e));
        m->arena_chunks.push_back(p); m->arena_off = 0; m->arena_cap = cap;
    }
    *out = (char*)m->arena_chunks.back() + m->arena_off;
    m->arena_off += need;
    HIPCHK(ctx, hipMemsetAsync(*out, 0, bytes, ctx->stream));
    return THK_OK;
}
static void free_seq(SeqBuf& s) {
    if (s.exec) hipGraphExecDestroy(s.exec);
    if (s.graph) hipGraphDestroy(s.graph);
    for (auto& g : s.multi) { if (g.second.second) hipGraphExecDestroy(g.second.second); if (g.second.first) hipGraphDestroy(g.second.first); }
    s.multi.clear(); s.multi_used.clear();
    hipFree(s.eng_ops);
    hipFree(s.kv);            // st, gen_log, hidden_in/out, logits, advance live in the model's arena
    s = SeqBuf();
}
static void free_working(thk_model* m) {
    for (auto& s : m->seqs) free_seq(s);
    m->seqs.clear();
    for (void* c : m->arena_chunks) hipFree(c);     // x, q, u, attn_out, part_*, block_best, rope_tab, counters, engine words, per-sequence state
    m->arena_chunks.clear(); m->arena_off = m->arena_cap = 0;
    hipFree(m->prefill_ws); hipFree(m->prefill_pk); m->prefill_pk = nullptr; m->prefill_pk_bytes = 0; m->pk_w.clear(); m->pk_tiles[0] = 0; m->pk_failed = false;
    m->eng_trace = nullptr;
    m->eng_gran = nullptr; m->eng_words = nullptr; m->engine = 0;
    m->x = m->q = m->u = m->attn_out = m->part_o = m->part_ml = nullptr; m->block_best = m->block_best_aux = nullptr; m->rope_tab = nullptr;
    m->prefill_ws = nullptr; m->prefill_ws_bytes = 0;
    m->finalized = false;
}
extern "C" int thk_model_destroy(thk_model* m) {
    if (!m) return THK_OK;
    hipSetDevice(m->ctx->device);
    hipStreamSynchronize(m->ctx->stream);
    free_working(m);
    hipFree(m->trace_buf);
    hipFree(m->weights_slab);     // every weight pointer of the stage points into it
    delete m;
    return THK_OK;
}
extern "C" int thk_model_uses_engine(const thk_model* m) { return (m && m->finalized && m->engine) ? 1 : 0; }
extern "C" int32_t thk_model_n_ff(const thk_model* m) { return m ? m->n_ff : 0; }
extern "C" int32_t thk_model_n_embd(const thk_model* m) { return m ? m->hp.n_embd : 0; }
extern "C" int32_t thk_model_n_ctx(const thk_model* m) { return m ? m->hp.n_ctx : 0; }

// name -> device slot; returns 0 ok, 1 = tensor belongs to another stage (ignored), <0 error
static int tensor_slot(thk_model* m, const char* name, void** dst, int64_t* ne0, int64_t* ne1, int* dtype) {
    const int64_t E = m->hp.n_embd, F = m->n_ff, V = m->hp.n_vocab;
    *dst = nullptr;
    if (!strcmp(name, "tok_embeddings.weight")) { *ne0 = E; *ne1 = V; *dtype = THK_F16; *dst = m->tok_embeddings; return *dst ? 0 : 1; }
    if (!strcmp(name, "norm.weight")) { *ne0 = E; *ne1 = 1; *dtype = THK_F32; *dst = m->norm; return *dst ? 0 : 1; }
    if (!strcmp(name, "output.weight")) { *ne0 = E; *ne1 = V; *dtype = THK_F16; *dst = m->output; return *dst ? 0 : 1; }
    int l = -1; char rest[64];
    if (sscanf(name, "layers.%d.%63s", &l, rest) != 2 || l < 0 || l >= m->hp.n_layer) return THK_ERR_NOTFOUND;
    const bool local = l >= m->l0 && l < m->l1;
    LayerW dummy; LayerW& L = local ? m->layers[l - m->l0] : dummy;
    struct { const char* n; void* p; int64_t c, r; int t; } tab[] = {
        {"attention_norm.weight", L.attention_norm, E, 1, THK_F32}, {"ffn_norm.weight", L.ffn_norm, E, 1, THK_F32},
        {"attention.wq.weight", L.wq, E, E, THK_F16}, {"attention.wk.weight", L.wk, E, E, THK_F16},
        {"attention.wv.weight", L.wv, E, E, THK_F16}, {"attention.wo.weight", L.wo, E, E, THK_F16},
        {"feed_forward.w1.weight", L.w1, E, F, THK_F16}, {"feed_forward.w2.weight", L.w2, F, E, THK_F16},
        {"feed_forward.w3.weight", L.w3, E, F, THK_F16}};
    for (auto& t : tab)
        if (!strcmp(rest, t.n)) { *ne0 = t.c; *ne1 = t.r; *dtype = t.t; *dst = t.p; return local ? 0 : 1; }
    return THK_ERR_NOTFOUND;
}
extern "C" int thk_model_set_tensor(thk_model* m, const char* name, int dtype, int64_t ne0, int64_t ne1, const void* host) {
    if (!m || !name || !host) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    void* dst; int64_t c, r; int t;
    const int rc = tensor_slot(m, name, &dst, &c, &r, &t);
    if (rc < 0) return fail(ctx, THK_ERR_NOTFOUND, "unknown tensor '%s'", name);
    if (ne1 <= 0) ne1 = 1;
    REQUIRE(ctx, c == ne0 && r == ne1, "tensor '%s': shape [%lld,%lld] expected [%lld,%lld]", name, (long long)ne1, (long long)ne0, (long long)r, (long long)c);
    REQUIRE(ctx, t == dtype, "tensor '%s': dtype %d expected %d (only GGML f16 models are supported, README.md:5)", name, dtype, t);
    if (rc == 1) return THK_OK;   // another stage owns it
    HIPCHK(ctx, hipSetDevice(ctx->device));
    m->pk_tiles[0] = 0; m->pk_w.clear(); m->pk_failed = false;   // the prefill tile images are stale now (and worth another try if they did not fit)
    HIPCHK(ctx, hipMemcpyAsync(dst, host, (size_t)(c * r) * (t == THK_F16 ? 2 : 4), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return THK_OK;
}
// Same as thk_model_set_tensor with the payload already in device memory (a thk_buf the caller uploaded, e.g. the host
// layer's TensorBuffer): one stream-ordered device-to-device copy into the model's slot.
extern "C" int thk_model_set_tensor_dev(thk_model* m, const char* name, int dtype, int64_t ne0, int64_t ne1, const void* dev_ptr) {
    if (!m || !name || !dev_ptr) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    void* dst; int64_t c, r; int t;
    const int rc = tensor_slot(m, name, &dst, &c, &r, &t);
    if (rc < 0) return fail(ctx, THK_ERR_NOTFOUND, "unknown tensor '%s'", name);
    if (ne1 <= 0) ne1 = 1;
    REQUIRE(ctx, c == ne0 && r == ne1, "tensor '%s': shape [%lld,%lld] expected [%lld,%lld]", name, (long long)ne1, (long long)ne0, (long long)r, (long long)c);
    REQUIRE(ctx, t == dtype, "tensor '%s': dtype %d expected %d (only GGML f16 models are supported, README.md:5)", name, dtype, t);
    if (rc == 1) return THK_OK;   // another stage owns it
    HIPCHK(ctx, hipSetDevice(ctx->device));
    m->pk_tiles[0] = 0; m->pk_w.clear(); m->pk_failed = false;
    HIPCHK(ctx, hipMemcpyAsync(dst, dev_ptr, (size_t)(c * r) * (t == THK_F16 ? 2 : 4), hipMemcpyDeviceToDevice, ctx->stream));
    return THK_OK;
}
// Read a tensor (or a byte range of it) back from the model's slot: the inverse of thk_model_set_tensor, for loaders' self-checks.
extern "C" int thk_model_get_tensor(thk_model* m, const char* name, int64_t offset_bytes, int64_t n_bytes, void* host_out) {
    if (!m || !name || !host_out) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    void* src; int64_t c, r; int t;
    const int rc = tensor_slot(m, name, &src, &c, &r, &t);
    if (rc < 0) return fail(ctx, THK_ERR_NOTFOUND, "unknown tensor '%s'", name);
    REQUIRE(ctx, rc == 0, "tensor '%s' belongs to another stage", name);
    const int64_t total = c * r * (t == THK_F16 ? 2 : 4);
    REQUIRE(ctx, offset_bytes >= 0 && n_bytes >= 0 && offset_bytes <= total && n_bytes <= total - offset_bytes, "tensor '%s': bytes [%lld, +%lld) outside its %lld bytes",
            name, (long long)offset_bytes, (long long)n_bytes, (long long)total);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(host_out, (const char*)src + offset_bytes, (size_t)n_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return THK_OK;
}
extern "C" int thk_model_fill_synthetic(thk_model* m, uint64_t seed, float sigma) {
    if (!m) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    const size_t E = m->hp.n_embd, F = m->n_ff, V = m->hp.n_vocab;
    const float sc = synth_scale(sigma);
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    m->pk_tiles[0] = 0; m->pk_w.clear(); m->pk_failed = false;
    if (m->tok_embeddings) HIPCHK(ctx, launch_synth_f16(synth_key("tok_embeddings.weight", seed), sc, V * E, m->tok_embeddings, st));
    if (m->norm) HIPCHK(ctx, launch_synth_gain(synth_key("norm.weight", seed), sc, E, m->norm, st));
    if (m->output) HIPCHK(ctx, launch_synth_f16(synth_key("output.weight", seed), sc, V * E, m->output, st));
    for (int l = m->l0; l < m->l1; ++l) {
        LayerW& L = m->layers[l - m->l0];
        char nm[96];
#define NM(s) (snprintf(nm, sizeof nm, "layers.%d." s, l), synth_key(nm, seed))
        HIPCHK(ctx, launch_synth_gain(NM("attention_norm.weight"), sc, E, L.attention_norm, st));
        HIPCHK(ctx, launch_synth_gain(NM("ffn_norm.weight"), sc, E, L.ffn_norm, st));
        HIPCHK(ctx, launch_synth_f16(NM("attention.wq.weight"), sc, E * E, L.wq, st));
        HIPCHK(ctx, launch_synth_f16(NM("attention.wk.weight"), sc, E * E, L.wk, st));
        HIPCHK(ctx, launch_synth_f16(NM("attention.wv.weight"), sc, E * E, L.wv, st));
        HIPCHK(ctx, launch_synth_f16(NM("attention.wo.weight"), sc, E * E, L.wo, st));
        HIPCHK(ctx, launch_synth_f16(NM("feed_forward.w1.weight"), sc, F * E, L.w1, st));
        HIPCHK(ctx, launch_synth_f16(NM("feed_forward.w2.weight"), sc, E * F, L.w2, st));
        HIPCHK(ctx, launch_synth_f16(NM("feed_forward.w3.weight"), sc, F * E, L.w3, st));
#undef NM
    }
    HIPCHK(ctx, hipStreamSynchronize(st));
    return THK_OK;
}
extern "C" int thk_model_set_lmhead_mode(thk_model* m, int mode) {
    if (!m) return THK_ERR_INVALID;
    REQUIRE(m->ctx, mode == THK_LMHEAD_CORRECT || mode == THK_LMHEAD_FAITHFUL, "bad lm-head mode %d", mode);
    REQUIRE(m->ctx, !m->finalized, "set the lm-head mode before thk_model_finalize (it is baked into the captured graph)");
    m->lm_mode = mode;
    return THK_OK;
}

__global__ void set_seq_state_kernel(SeqState* st, int token, int pos, int reset_gen) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { st->token = token; st->pos = pos; if (reset_gen) { st->n_gen = 0; st->pad = 0; } }
}
__global__ void set_seq_token_kernel(SeqState* st, int token) {
    if (threadIdx.x == 0 && blockIdx.x == 0) st->token = token;
}
int set_seq_state(thk_model* m, int seq, int token, int pos, bool reset_gen) {
    m->seqs[seq].pos_host = pos;
    hipLaunchKernelGGL(set_seq_state_kernel, dim3(1), dim3(64), 0, m->ctx->stream, m->seqs[seq].st, token, pos, reset_gen ? 1 : 0);
    HIPCHK(m->ctx, hipGetLastError());
    return THK_OK;
}
extern "C" int thk_model_finalize(thk_model* m) {
    if (!m) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    free_working(m);
    const size_t E = m->hp.n_embd, H = m->hp.n_head, D = E / H, F = m->n_ff, V = m->hp.n_vocab, T = m->hp.n_ctx;
    const int nl = m->l1 - m->l0;
    // launch geometry
    m->nsplit = (int)tun(ctx, "attn_splits");
    if (m->nsplit == 0) m->nsplit = T > 512 ? 8 : 4;        // auto: 4 context splits per head, 8 for caches longer than 512 rows (round 6: a 1024-row cache in 8 splits is ONE round of an
                                                            // 8-wave workgroup - same-box A/B 2.4495 -> 2.409 ms per step at T = 1024; round 4 switched at 1024 rows)
    REQUIRE(ctx, valid_splits(m->nsplit), "attn_splits must be 0 (auto), 1, 2, 4 or 8");
    m->tc = (int)((T + m->nsplit - 1) / m->nsplit);
    m->nt = true;
    m->use_graph = tun(ctx, "use_graph") != 0;
    m->skip_kernel = (int)tun(ctx, "measure_skip_kernel");
    m->kv_f16 = tun(ctx, "kv_f16") != 0;
    {   // waves per attention workgroup.  0 = auto: 8 - or 16 for an f32 cache longer than 1024 rows (D = 128): a split of 256 positions is then ONE round of the workgroup,
        // every wave issues its single K/V batch at once (round 6, same-box A/B at T = 2048: step 2.6506 -> 2.6163 ms; the binary16 cache and T = 512 are faster with 8)
        const int64_t aw = tun(ctx, "attn_waves");
        REQUIRE(ctx, aw == 0 || aw == 4 || aw == 8 || aw == 16, "attn_waves must be 0 (auto), 4, 8 or 16");
        REQUIRE(ctx, aw != 16 || D == 128, "attn_waves = 16 exists for head dimension 128 only");
        m->attn_waves = aw != 0 ? (int)aw : ((T > 1024 && !m->kv_f16 && D == 128) ? 16 : 8);
    }
    m->fold_embed = tun(ctx, "fold_embed") != 0;
    m->fold_finish = (int)tun(ctx, "fold_finish");   // 0 | 1 | 2
    REQUIRE(ctx, m->fold_finish >= 0 && m->fold_finish <= 2, "fold_finish must be 0, 1 or 2");
    m->attn_tc_dyn = tun(ctx, "attn_tc_dyn") != 0;
    m->gain_alias = tun(ctx, "measure_gain_alias") != 0;
    m->var_qkv = resolve_variant(ctx, "qkv", (int)E); m->var_wo = resolve_variant(ctx, "wo", (int)E);
    m->var_w13 = resolve_variant(ctx, "w13", (int)E); m->var_w2 = resolve_variant(ctx, "w2", (int)E); m->var_head = resolve_variant(ctx, "head", (int)E);
    m->grid_qkv = grid_for(ctx, "gemv_bpc_qkv", (int)(3 * E / gemv_rows_per_group((int)E, GEMV_EPI_ROPE_KV, m->var_qkv)), (int)E);     // row pairs, or single rows (variants 1, 6)
    // the single-row forms keep their pair sums in 32 LDS rounds per wave (gemv_body): a geometry with more rows per wave than that
    // (few CUs, a small explicit grid) takes the row-pair form instead of failing at the first launch
    auto fits_single_rows = [&](int var, int epi, int64_t rows, int grid) { return gemv_rows_per_group((int)E, epi, var) != 1 || (int64_t)grid * kWaves * 32 >= rows; };
    if (!fits_single_rows(m->var_qkv, GEMV_EPI_ROPE_KV, 3 * E, m->grid_qkv)) { m->var_qkv = 5; m->grid_qkv = grid_for(ctx, "gemv_bpc_qkv", (int)(3 * E / 2), (int)E); }
    m->grid_wo = grid_for(ctx, "gemv_bpc_wo", (int)((E + gemv_rows_per_group((int)E, GEMV_EPI_RESID, m->var_wo) - 1) / gemv_rows_per_group((int)E, GEMV_EPI_RESID, m->var_wo)), (int)E);
    m->grid_w13 = grid_for(ctx, "gemv_bpc_w13", (int)(2 * F / gemv_rows_per_group((int)E, GEMV_EPI_SWIGLU, m->var_w13)), (int)E);     // (w1, w3) row pairs, or single rows (variants 1, 6)
    if (!fits_single_rows(m->var_w13, GEMV_EPI_SWIGLU, 2 * F, m->grid_w13)) { m->var_w13 = 5; m->grid_w13 = grid_for(ctx, "gemv_bpc_w13", (int)F, (int)E); }
    m->grid_w2 = grid_for(ctx, "gemv_bpc_w2", (int)((E + gemv_rows_per_group((int)F, GEMV_EPI_RESID, m->var_w2) - 1) / gemv_rows_per_group((int)F, GEMV_EPI_RESID, m->var_w2)), (int)E);
    m->grid_head = grid_for(ctx, "gemv_bpc_head", (int)((V + gemv_rows_per_group((int)E, GEMV_EPI_HEAD, m->var_head) - 1) / gemv_rows_per_group((int)E, GEMV_EPI_HEAD, m->var_head)), (int)E);
    // working buffers
#define ALLOCZ(ptr, bytes)                                                                                              \
    do { const int rc_ = arena_alloc(m, (void**)&(ptr), (bytes)); if (rc_ != THK_OK) return rc_; } while (0)
#define ALLOCZ_OWN(ptr, bytes)                                                                                          \
    do {                                                                                                                \
        hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                                                             \
        if (e_ != hipSuccess) return fail(ctx, e_ == hipErrorOutOfMemory ? THK_ERR_OOM : THK_ERR_HIP, "hipMalloc(%zu) for %s: %s", (size_t)(bytes), #ptr, hipGetErrorString(e_)); \
        HIPCHK(ctx, hipMemsetAsync((ptr), 0, (bytes), ctx->stream));                                                    \
    } while (0)
    ALLOCZ(m->x, E * 4); ALLOCZ(m->q, E * 4); ALLOCZ(m->u, F * 4); ALLOCZ(m->attn_out, E * 4);
    ALLOCZ(m->part_o, H * kMaxSplit * D * 4); ALLOCZ(m->part_ml, H * kMaxSplit * 2 * 4);
    {   // arg-max key slots, twice: [0] the decode step's (all zero between launches when the pick is folded into the lm-head launch),
        // [1] for head launches outside the step (prefill), whose keys nobody consumes
        const size_t slots = (size_t)std::max(m->grid_head > 0 ? m->grid_head : 1, ctx->n_cu) + 512;
        ALLOCZ(m->block_best, slots * 8 * 2);
        m->block_best_aux = m->block_best + slots;
        m->block_best_slots = slots;
    }
    ALLOCZ(m->rope_tab, T * (D / 2) * 2 * 4);
    {
        std::vector<float> tab;
        build_rope_table(tab, (int)D, 0, (int)T);
        HIPCHK(ctx, hipMemcpyAsync(m->rope_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    m->seqs.resize(m->n_seq);
    for (auto& s : m->seqs) {
        ALLOCZ_OWN(s.kv, (size_t)nl * 2 * T * E * (m->kv_f16 ? 2 : 4));
        ALLOCZ(s.st, sizeof(SeqState)); ALLOCZ(s.gen_log, (size_t)kGenLogCap * 4); ALLOCZ(s.clock_log, (size_t)kGenLogCap * 8);
        ALLOCZ(s.hidden_in, E * 4); ALLOCZ(s.hidden_out, E * 4); ALLOCZ(s.advance, 4);
        if (m->flags & THK_STAGE_HEAD) ALLOCZ(s.logits, V * 4);
        s.advance_host = 0;
    }
    m->engine = engine_plan(m) ? 1 : 0;
    if (m->engine) {
        const size_t ngran = 2 * E + 3 * E + E + F + H * (size_t)m->eng_nsplit * (D + 2);
        ALLOCZ(m->eng_gran, ngran * 8);
        ALLOCZ(m->eng_words, 64 * 4);
        HIPCHK(ctx, hipMemsetAsync(m->eng_words, 0, 4, ctx->stream));
        unsigned one = 1;                                   // epoch 0 would make a zero-filled granule look valid for op tag 0
        HIPCHK(ctx, hipMemcpyAsync(m->eng_words, &one, 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (auto& sq : m->seqs) { int rc_ = engine_build_program(m, sq); if (rc_ != THK_OK) return rc_; }
        if (tun(ctx, "engine_trace") != 0) ALLOCZ(m->eng_trace, (size_t)ctx->n_cu * m->seqs[0].eng_n_ops * 8 * 8);
    }
#undef ALLOCZ
#undef ALLOCZ_OWN
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    m->finalized = true;
    // warm-up (loads code objects, sets LDS attributes) then capture one graph per sequence
    int rc = step_enqueue(m, 0);
    if (rc != THK_OK) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    rc = thk_model_reset_kv(m, 0);
    if (rc != THK_OK) return rc;
    if (m->use_graph) {
        for (int s = 0; s < m->n_seq; ++s) {
            HIPCHK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
            rc = step_enqueue(m, s);
            hipError_t e = hipStreamEndCapture(ctx->stream, &m->seqs[s].graph);
            if (rc != THK_OK) return rc;
            if (e != hipSuccess) return fail(ctx, THK_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
            HIPCHK(ctx, hipGraphInstantiate(&m->seqs[s].exec, m->seqs[s].graph, nullptr, nullptr, 0));
        }
    }
    return THK_OK;
}

extern "C" int thk_model_reset_kv(thk_model* m, int32_t seq) {
    if (!m) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    REQUIRE(ctx, m->finalized, "thk_model_reset_kv before thk_model_finalize");
    REQUIRE(ctx, seq >= 0 && seq < m->n_seq, "bad sequence %d", seq);
    const size_t bytes = (size_t)(m->l1 - m->l0) * 2 * m->hp.n_ctx * m->hp.n_embd * (m->kv_f16 ? 2 : 4);
    HIPCHK(ctx, hipMemsetAsync(m->seqs[seq].kv, 0, bytes, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(m->seqs[seq].st, 0, sizeof(SeqState), ctx->stream));
    m->seqs[seq].pos_host = 0;
    return THK_OK;
}

extern "C" int thk_model_eval(thk_model* m, int32_t seq, const int32_t* tokens, int32_t n_tokens, int32_t n_past,
                              float* hidden_inout, float* logits_out) {
    if (!m) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    REQUIRE(ctx, m->finalized, "thk_model_eval before thk_model_finalize");
    REQUIRE(ctx, seq >= 0 && seq < m->n_seq, "bad sequence %d", seq);
    REQUIRE(ctx, n_tokens >= 1 && n_past >= 0 && n_past + n_tokens <= m->hp.n_ctx, "n_past=%d + n_tokens=%d exceeds n_ctx=%d", n_past, n_tokens, m->hp.n_ctx);
    const bool embed = m->flags & THK_STAGE_EMBED, head = m->flags & THK_STAGE_HEAD;
    REQUIRE(ctx, !embed || tokens, "an embedding stage needs token ids");
    REQUIRE(ctx, embed || (hidden_inout && n_tokens == 1), "a non-embedding stage takes exactly one hidden state per call");
    REQUIRE(ctx, !logits_out || head, "logits requested from a stage without the lm-head");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    SeqBuf& sb = m->seqs[seq];
    const size_t E = m->hp.n_embd, V = m->hp.n_vocab;
    if (!embed) HIPCHK(ctx, hipMemcpyAsync(sb.hidden_in, hidden_inout, E * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = step_set_advance(m, seq, 0);
    if (rc != THK_OK) return rc;
    for (int i = 0; i < n_tokens; ++i) {
        if (embed) REQUIRE(ctx, tokens[i] >= 0 && tokens[i] < m->hp.n_vocab, "token id %d out of range", tokens[i]);
        rc = set_seq_state(m, seq, embed ? tokens[i] : 0, n_past + i, false);
        if (rc != THK_OK) return rc;
        rc = step_run(m, seq);
        if (rc != THK_OK) return rc;
    }
    if (logits_out) HIPCHK(ctx, hipMemcpyAsync(logits_out, sb.logits, V * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (hidden_inout) HIPCHK(ctx, hipMemcpyAsync(hidden_inout, head ? m->x : sb.hidden_out, E * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return check_engine_error(m);
}

// th_eval_gpu + the candidate selection of a STOCHASTIC sampler in one stream round trip: the step(s), then the top-k kernel behind them on the
// same stream, its k keys written straight into host-mapped memory and a stamp behind them that the host thread polls (no stream synchronisation).  (thk_model_eval + thk_model_logits_topk cost two
// synchronisations, a launch from an idle stream and a copy operation per token: 96 us next to a 2.27 ms step; this form 4 x less.)
extern "C" int thk_model_eval_topk(thk_model* m, int32_t seq, const int32_t* tokens, int32_t n_tokens, int32_t n_past, int32_t k, float* values_out, int32_t* ids_out) {
    if (!m || !tokens || !values_out || !ids_out) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    REQUIRE(ctx, m->finalized, "thk_model_eval_topk before thk_model_finalize");
    REQUIRE(ctx, seq >= 0 && seq < m->n_seq, "bad sequence %d", seq);
    REQUIRE(ctx, (m->flags & THK_STAGE_EMBED) && (m->flags & THK_STAGE_HEAD), "thk_model_eval_topk needs a full-model stage (embedding + head)");
    REQUIRE(ctx, n_tokens >= 1 && n_past >= 0 && n_past + n_tokens <= m->hp.n_ctx, "n_past=%d + n_tokens=%d exceeds n_ctx=%d", n_past, n_tokens, m->hp.n_ctx);
    REQUIRE(ctx, k >= 1 && k <= 1024 && k <= m->hp.n_vocab && m->hp.n_vocab <= 32768, "top-k: k=%d / n_vocab=%d outside the device kernel's range (n_vocab <= 32768, k <= 1024)", k, m->hp.n_vocab);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = step_set_advance(m, seq, 0);
    if (rc != THK_OK) return rc;
    for (int i = 0; i < n_tokens; ++i) {
        REQUIRE(ctx, tokens[i] >= 0 && tokens[i] < m->hp.n_vocab, "token id %d out of range", tokens[i]);
        if ((rc = set_seq_state(m, seq, tokens[i], n_past + i, false)) != THK_OK) return rc;
        if ((rc = step_run(m, seq)) != THK_OK) return rc;
    }
    if ((rc = topk_enqueue_pinned(ctx, m->seqs[seq].logits, m->hp.n_vocab, k)) != THK_OK) return rc;
    if ((rc = topk_wait_pinned(ctx)) != THK_OK) return rc;
    topk_decode_keys(ctx->pinned_keys, k, values_out, ids_out);
    return m->engine ? check_engine_error(m) : THK_OK;
}

extern "C" int thk_model_seq_set(thk_model* m, int32_t seq, int32_t token, int32_t pos) {
    if (!m) return THK_ERR_INVALID;
    REQUIRE(m->ctx, m->finalized && seq >= 0 && seq < m->n_seq, "bad sequence %d (or model not finalized)", seq);
    REQUIRE(m->ctx, pos >= 0 && pos < m->hp.n_ctx && token >= 0 && token < m->hp.n_vocab, "token %d / pos %d out of range", token, pos);
    HIPCHK(m->ctx, hipSetDevice(m->ctx->device));      // a host worker thread (capi_on_human_message) starts with device 0 current
    return set_seq_state(m, seq, token, pos, true);
}
extern "C" int thk_model_seq_set_token(thk_model* m, int32_t seq, int32_t token) {
    if (!m) return THK_ERR_INVALID;
    REQUIRE(m->ctx, m->finalized && seq >= 0 && seq < m->n_seq, "bad sequence %d (or model not finalized)", seq);
    REQUIRE(m->ctx, token >= 0 && token < m->hp.n_vocab, "token %d out of range", token);
    HIPCHK(m->ctx, hipSetDevice(m->ctx->device));
    hipLaunchKernelGGL(set_seq_token_kernel, dim3(1), dim3(64), 0, m->ctx->stream, m->seqs[seq].st, token);
    HIPCHK(m->ctx, hipGetLastError());
    return THK_OK;
}
extern "C" void* thk_model_hidden_in(thk_model* m, int32_t seq) { return (m && m->finalized && seq >= 0 && seq < m->n_seq) ? m->seqs[seq].hidden_in : nullptr; }
extern "C" void* thk_model_hidden_out(thk_model* m, int32_t seq) { return (m && m->finalized && seq >= 0 && seq < m->n_seq) ? m->seqs[seq].hidden_out : nullptr; }
extern "C" void* thk_model_token_dev(thk_model* m, int32_t seq) { return (m && m->finalized && seq >= 0 && seq < m->n_seq) ? (void*)&m->seqs[seq].st->token : nullptr; }
extern "C" void* thk_model_logits_dev(thk_model* m, int32_t seq) { return (m && m->finalized && seq >= 0 && seq < m->n_seq) ? m->seqs[seq].logits : nullptr; }

// The folded greedy pick gave up on a key slot (a workgroup of an lm-head launch never delivered within 1 s of device time): the
// finisher zeroed the slot and set SeqState::pad, but the late workgroup's write-through key may have landed AFTERWARDS - a non-zero
// slot at the start of the next launch, i.e. a stale key that every later step could reduce (advisor, round 4).  Nothing on the device
// repairs that, so the host does when it sees the word: drain the stream, zero every key slot, clear the word, report the failure once.
int report_pick_timeout(thk_model* m, int seq) {
    thk_ctx* ctx = m->ctx;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (m->block_best && m->block_best_slots) HIPCHK(ctx, hipMemsetAsync(m->block_best, 0, m->block_best_slots * 8, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(&m->seqs[seq].st->pad, 0, 4, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return fail(ctx, THK_ERR_STATE, "sequence %d: the lm-head launch's folded greedy pick gave up waiting for an arg-max key (a workgroup of the launch never delivered); "
                "tokens from that step on are not trustworthy - the key slots have been cleared, set the sequence again (thk_model_seq_set) before decoding on", seq);
}

extern "C" int thk_model_seq_get(thk_model* m, int32_t seq, int32_t* tokens_out, int32_t cap, int32_t* n_out, int32_t* pos_out) {
    if (!m) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    REQUIRE(ctx, m->finalized && seq >= 0 && seq < m->n_seq, "bad sequence %d (or model not finalized)", seq);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    SeqState h{};
    HIPCHK(ctx, hipMemcpyAsync(&h, m->seqs[seq].st, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    int n = h.n_gen < kGenLogCap ? h.n_gen : kGenLogCap;
    if (n > cap) n = cap;
    if (n > 0 && tokens_out) {
        HIPCHK(ctx, hipMemcpyAsync(tokens_out, m->seqs[seq].gen_log, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (n_out) *n_out = h.n_gen;
    if (pos_out) *pos_out = h.pos;
    if (h.pad != 0) return report_pick_timeout(m, seq);
    return check_engine_error(m);
}

// The logits of the sequence's last evaluated token, without the 4 * n_vocab-byte read-back: the k largest (value descending, ties
// by ascending id) selected on the device, or the whole vector on request.
extern "C" int thk_model_logits_topk(thk_model* m, int32_t seq, int32_t k, float* values_out, int32_t* ids_out) {
    if (!m) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    REQUIRE(ctx, m->finalized && seq >= 0 && seq < m->n_seq && (m->flags & THK_STAGE_HEAD), "bad sequence %d, model not finalized, or a stage without the lm-head", seq);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return topk_to_host(ctx, m->seqs[seq].logits, m->hp.n_vocab, k, values_out, ids_out);
}
extern "C" int thk_model_read_logits(thk_model* m, int32_t seq, float* logits_out) {
    if (!m || !logits_out) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    REQUIRE(ctx, m->finalized && seq >= 0 && seq < m->n_seq && (m->flags & THK_STAGE_HEAD), "bad sequence %d, model not finalized, or a stage without the lm-head", seq);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(logits_out, m->seqs[seq].logits, (size_t)m->hp.n_vocab * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return THK_OK;
}
// Device-side step clock: clock_out[i] = value of the chip-wide 100 MHz counter (s_memrealtime) when the step that logged token i
// (thk_model_seq_get) finished.  Differences of consecutive entries are per-step durations measured on the GPU itself, inside
// replayed multi-step graphs, with nothing inserted into the stream (bench.py's p5/p50/p95).
extern "C" int thk_model_seq_clock(thk_model* m, int32_t seq, unsigned long long* clock_out, int32_t cap, int32_t* n_out) {
    if (!m || !clock_out || !n_out) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    REQUIRE(ctx, m->finalized && seq >= 0 && seq < m->n_seq && (m->flags & THK_STAGE_HEAD), "bad sequence %d, model not finalized, or a stage without the lm-head", seq);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    SeqState h{};
    HIPCHK(ctx, hipMemcpyAsync(&h, m->seqs[seq].st, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    int n = h.n_gen < kGenLogCap ? h.n_gen : kGenLogCap;
    if (n > cap) n = cap;
    if (n > 0) {
        HIPCHK(ctx, hipMemcpyAsync(clock_out, m->seqs[seq].clock_log, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    *n_out = n;
    return THK_OK;
}
extern "C" int thk_model_seq_last_token(thk_model* m, int32_t seq, int32_t* token_out) {
    if (!m || !token_out) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    REQUIRE(ctx, m->finalized && seq >= 0 && seq < m->n_seq, "bad sequence %d (or model not finalized)", seq);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    SeqState h{};
    HIPCHK(ctx, hipMemcpyAsync(&h, m->seqs[seq].st, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *token_out = h.token;
    if (h.pad != 0) return report_pick_timeout(m, seq);
    return check_engine_error(m);
}

// Development aid: the working buffers a decode step leaves behind (the LAST layer's q, split partials and u, the final hidden
// state), so that two launch configurations can be compared stage by stage on a one-layer model.
extern "C" int thk_model_debug_buffer(thk_model* m, const char* name, float* out, int64_t cap, int64_t* n_out) {
    if (!m || !name || !out || !n_out) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    REQUIRE(ctx, m->finalized, "model not finalized");
    const size_t E = m->hp.n_embd, H = m->hp.n_head, D = E / H, F = m->n_ff;
    const float* p = nullptr; size_t n = 0;
    if (!strcmp(name, "x")) { p = m->x; n = E; }
    else if (!strcmp(name, "q")) { p = m->q; n = E; }
    else if (!strcmp(name, "u")) { p = m->u; n = F; }
    else if (!strcmp(name, "part_o")) { p = m->part_o; n = H * m->nsplit * D; }
    else if (!strcmp(name, "part_ml")) { p = m->part_ml; n = H * m->nsplit * 2; }
    else return fail(ctx, THK_ERR_NOTFOUND, "no working buffer '%s' (x, q, u, part_o, part_ml)", name);
    REQUIRE(ctx, (int64_t)n <= cap, "buffer '%s' holds %zu floats", name, n);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(out, p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = (int64_t)n;
    return THK_OK;
}

extern "C" int64_t thk_model_bytes_per_token(const thk_model* m, int32_t T) {
    if (!m) return 0;
    const int64_t E = m->hp.n_embd, F = m->n_ff, V = m->hp.n_vocab, nl = m->l1 - m->l0;
    const int64_t s_kv = m->kv_f16 ? 2 : 4;          // a build that stores KV as f16 must say so and use s_kv = 2
    int64_t b = nl * ((4 * E * E + 3 * E * F) * 2      // f16 weights
                      + 2 * (int64_t)T * E * s_kv       // K,V read
                      + 2 * E * s_kv                    // K,V row written
                      + 2 * E * 4);                     // two norm gains
    if (m->flags & THK_STAGE_HEAD) b += V * E * 2 + E * 4;
    return b;
}
