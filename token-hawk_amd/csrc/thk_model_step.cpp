// thk_model_step.cpp — one decode step of a stage: th_eval_gpu's body (th-llama.cpp:464-660; build_layer_cmdbuf :270-452,
// build_final_compute_cmdbuf :240-268) as 5 fused launches per layer + the lm-head launch that also picks the greedy token, enqueued
// eagerly or replayed from hipGraphs of up to 32 steps; the step-level API (thk_model_decode_step(s), thk_model_prepare_steps) and
// the two development views of a step (per-launch HIP-event times, per-wave device timestamps).
#include "thk_internal.hpp"

// ----- one decode step of this stage, enqueued on the ctx stream (eager or under capture)
struct StepProf {
    std::vector<std::string> names;
    std::vector<hipEvent_t> events;   // events[i] recorded before kernel i; one extra at the end
    bool names_only = false;          // collect the launch names without recording events (the step trace must not perturb the stream)
};
static int prof_mark(thk_ctx* ctx, StepProf* p, const char* name) {
    if (!p) return THK_OK;
    if (p->names_only) { if (name) p->names.push_back(name); return THK_OK; }
    hipEvent_t ev;
    HIPCHK(ctx, hipEventCreate(&ev));
    HIPCHK(ctx, hipEventRecord(ev, ctx->stream));
    p->events.push_back(ev);
    if (name) p->names.push_back(name);
    return THK_OK;
}
#define MARK(name) do { int rc_ = prof_mark(ctx, prof, name); if (rc_ != THK_OK) return rc_; } while (0)

static int enqueue_step(thk_model* m, int seq, StepProf* prof) {
    thk_ctx* ctx = m->ctx;
    hipStream_t st = ctx->stream;
    SeqBuf& sb = m->seqs[seq];
    const int E = m->hp.n_embd, H = m->hp.n_head, D = E / H, F = m->n_ff, V = m->hp.n_vocab, T = m->hp.n_ctx;
    const bool nt = m->nt != 0;
    const int nl = m->l1 - m->l0;
    const float* xin = sb.hidden_in;
    int trace_k = 0;                                     // development timeline (thk_model_step_trace): one [kTraceBlocks][8][4] slab per launch
    auto trace_slab = [&]() -> unsigned long long* { return m->trace_on ? m->trace_buf + (size_t)(trace_k++) * kTraceBlocks * kTraceWords : nullptr; };
    const bool fold_embed = (m->flags & THK_STAGE_EMBED) && m->fold_embed && !m->engine && nl > 0 && m->skip_kernel != 1;
    if (m->flags & THK_STAGE_EMBED) {
        if (!fold_embed) {      // with fold_embed the first layer's qkv prologue fetches the row itself (ProRms<.., EMB>)
            MARK("embed");
            HIPCHK(ctx, launch_embed(m->tok_embeddings, sb.st, 0, E, m->x, st, trace_slab()));
        }
        xin = m->x;
    }
    if (m->engine) {   // every layer (+ lm-head) of this stage in ONE persistent launch; the program was built at finalize
        EngArgs a{};
        a.ops = sb.eng_ops; a.n_ops = sb.eng_n_ops; a.st = sb.st; a.epoch = m->eng_words; a.err = m->eng_words + 32;
        a.E = E; a.H = H; a.D = D; a.nsplit = m->eng_nsplit; a.tc = m->eng_tc;
        a.NS = m->eng_NS; a.v0_bytes = m->eng_v0; a.v1_bytes = m->eng_v1;
        a.rope_tab = m->rope_tab; a.scale = 1.0f / sqrtf((float)D); a.block_best = m->block_best; a.trace = m->eng_trace;
        MARK("engine");
        HIPCHK(ctx, launch_engine(a, ctx->n_cu, st));
        if (m->flags & THK_STAGE_HEAD) {
            MARK("finish_token");
            HIPCHK(ctx, launch_finish_token(m->block_best, ctx->n_cu, sb.st, sb.gen_log, kGenLogCap, sb.advance, nullptr, T, m->eng_words, st, nullptr, sb.clock_log));
        } else {
            MARK("advance_pos");
            HIPCHK(ctx, launch_advance_pos(sb.st, sb.advance, T, m->eng_words, st));
        }
        MARK(nullptr);
        return THK_OK;
    }
    for (int i = 0; i < nl; ++i) {
        const LayerW& L = m->layers[i];
        float* kc = kcache_of(m, sb, i);
        float* vc = vcache_of(m, sb, i);
        const float* xr_in = i == 0 ? xin : m->x;
        {   // rms_norm*gain -> wq,wk,wv -> RoPE -> K/V append   (steps 1-4, th-llama.cpp:299-339)
            GemvArgs a{};
            a.W[0] = L.wq; a.W[1] = L.wk; a.W[2] = L.wv; a.R = E; a.C = E; a.n_groups = 3 * E / gemv_rows_per_group(E, GEMV_EPI_ROPE_KV, m->var_qkv);
            a.x = xr_in; a.gain = L.attention_norm; a.y = m->q;
            a.kcache = kc; a.vcache = vc; a.rope_tab = m->rope_tab; a.pos_ptr = &sb.st->pos; a.E = E; a.D = D; a.kv_f16 = m->kv_f16;
            const bool emb = fold_embed && i == 0;
            if (emb) { a.embed = m->tok_embeddings; a.tok_ptr = &sb.st->token; a.x_out = m->x; }
            if (m->gain_alias && !emb) a.gain = a.x;
            a.trace = trace_slab();
            MARK("norm_qkv_rope_kv");
            if (m->skip_kernel != 1) HIPCHK(ctx, launch_gemv(emb ? GEMV_PRO_RMS_EMBED : GEMV_PRO_RMS, GEMV_EPI_ROPE_KV, m->var_qkv, a, m->grid_qkv, nt, st));
        }
        {   // attention over the cache in place (steps 5-9, th-llama.cpp:341-397), then
            // split combine -> wo -> + residual (steps 10-11, th-llama.cpp:401-413)
            AttnArgs t{};
            t.q = m->q; t.kcache = kc; t.vcache = vc; t.pos_ptr = &sb.st->pos; t.H = H; t.D = D; t.nsplit = m->nsplit; t.tc = m->tc;
            t.tc_dyn = m->attn_tc_dyn;
            t.pipe = (T + m->nsplit - 1) / m->nsplit > attn_round_positions(D, m->attn_waves, m->kv_f16 != 0);      // a split of the full cache is longer than one round
            t.scale = 1.0f / sqrtf((float)D); t.waves = m->attn_waves; t.kv_f16 = m->kv_f16;
            t.out = m->nsplit == 1 ? m->attn_out : nullptr; t.part_o = m->part_o; t.part_ml = m->part_ml;
            GemvArgs a{};
            a.W[0] = L.wo; a.R = E; a.C = E;
            const int NR = gemv_rows_per_group(E, GEMV_EPI_RESID, m->var_wo);
            a.n_groups = (E + NR - 1) / NR;
            a.x = m->attn_out; a.part_o = m->part_o; a.part_ml = m->part_ml; a.H = H; a.D = D; a.nsplit = m->nsplit;
            a.resid = xr_in; a.y = m->x;
            t.trace = trace_slab();
            MARK("attn_decode");
            if (m->skip_kernel != 2) HIPCHK(ctx, launch_attn_decode(t, st));
            a.trace = trace_slab();
            MARK("attn_wo_resid");
            if (m->skip_kernel != 3) HIPCHK(ctx, launch_gemv(m->nsplit == 1 ? GEMV_PRO_COPY : GEMV_PRO_ATTN, GEMV_EPI_RESID, m->var_wo, a, m->grid_wo, nt, st));
        }
        {   // rms_norm*gain -> w1,w3 -> silu*gate   (steps 12-14, th-llama.cpp:415-438)
            GemvArgs a{};
            a.W[0] = L.w1; a.W[1] = L.w3; a.R = F; a.C = E; a.n_groups = 2 * F / gemv_rows_per_group(E, GEMV_EPI_SWIGLU, m->var_w13);
            a.x = m->x; a.gain = L.ffn_norm; a.y = m->u;
            if (m->gain_alias) a.gain = a.x;
            a.trace = trace_slab();
            MARK("norm_w13_swiglu");
            if (m->skip_kernel != 4) HIPCHK(ctx, launch_gemv(GEMV_PRO_RMS, GEMV_EPI_SWIGLU, m->var_w13, a, m->grid_w13, nt, st));
        }
        {   // w2 -> + residual   (steps 15-16, th-llama.cpp:440-451)
            GemvArgs a{};
            a.W[0] = L.w2; a.R = E; a.C = F;
            const int NR = gemv_rows_per_group(F, GEMV_EPI_RESID, m->var_w2);
            a.n_groups = (E + NR - 1) / NR;
            a.x = m->u; a.resid = m->x;
            a.y = (i == nl - 1 && !(m->flags & THK_STAGE_HEAD)) ? sb.hidden_out : m->x;
            a.trace = trace_slab();
            MARK("w2_resid");
            if (m->skip_kernel != 5) {
                if (m->var_w2 >= 8 && gemv_quarter_ok(F, E, m->grid_w2)) HIPCHK(ctx, launch_gemv_quarter(a, m->grid_w2, st));     // a workgroup per row (variants 8, 9)
                else HIPCHK(ctx, launch_gemv(GEMV_PRO_COPY, GEMV_EPI_RESID, m->var_w2, a, m->grid_w2, nt, st));
            }
        }
    }
    if (m->flags & THK_STAGE_HEAD) {   // final norm -> lm-head -> greedy pick   (th-llama.cpp:240-268, :826-838)
        GemvArgs a{};
        a.W[0] = m->output; a.R = V; a.C = E;
        const int NR = gemv_rows_per_group(E, GEMV_EPI_HEAD, m->var_head);
        a.n_groups = (V + NR - 1) / NR;
        a.x = m->x; a.gain = m->norm; a.y = sb.logits;
        a.lm_faithful = m->lm_mode == THK_LMHEAD_FAITHFUL; q1_constants(V, &a.q1_split, &a.q1_cov);
        a.block_best = m->block_best;
        a.trace = trace_slab();
        FinishArgs f{};
        f.block_best = m->block_best; f.nblocks = m->grid_head; f.st = sb.st; f.gen_log = sb.gen_log; f.log_cap = kGenLogCap; f.advance_ptr = sb.advance;
        f.n_ctx = T; f.clock_log = sb.clock_log;
        const bool fold = m->fold_finish && m->skip_kernel != 6;
        if (fold) { a.fin = f; a.fin.folded = m->fold_finish; }      // one workgroup of the launch (1: the highest-numbered, 2: workgroup 0) picks the token itself (key slots are zero between launches)
        MARK("norm_lmhead");
        if (m->skip_kernel != 6) HIPCHK(ctx, launch_gemv(GEMV_PRO_RMS, GEMV_EPI_HEAD, m->var_head, a, m->grid_head, nt, st));
        if (!fold) {
            MARK("finish_token");
            if (m->skip_kernel != 6) { f.trace = trace_slab(); HIPCHK(ctx, launch_finish_token_args(f, st)); }
            else HIPCHK(ctx, launch_advance_pos(sb.st, sb.advance, T, nullptr, st));       // no arg-max keys were written: keep the token
        }
    } else {
        MARK("advance_pos");
        HIPCHK(ctx, launch_advance_pos(sb.st, sb.advance, T, nullptr, st));
    }
    MARK(nullptr);
    return THK_OK;
}

static int set_advance(thk_model* m, int seq, int advance) {
    SeqBuf& sb = m->seqs[seq];
    advance = advance ? 1 : 0;
    if (sb.advance_host != advance) {
        HIPCHK(m->ctx, hipMemsetAsync(sb.advance, advance ? 1 : 0, 4, m->ctx->stream));
        sb.advance_host = advance;
    }
    return THK_OK;
}
static int run_step(thk_model* m, int seq) {
    if (m->use_graph && m->seqs[seq].exec) { HIPCHK(m->ctx, hipGraphLaunch(m->seqs[seq].exec, m->ctx->stream)); return THK_OK; }
    return enqueue_step(m, seq, nullptr);
}

// A step at position p evaluates T = p + 1 <= n_ctx cache rows and writes row p; an advancing step leaves p + 1.  The
// host mirrors the device position exactly (every change goes through this API), so running past the context is
// refused here instead of corrupting the caches (the device-side clamp in finish_token / advance_pos is the backstop).
static int check_room(thk_model* m, int seq, int n_steps, int advance) {
    const SeqBuf& sb = m->seqs[seq];
    const int last = sb.pos_host + (advance ? n_steps - 1 : 0);
    REQUIRE(m->ctx, n_steps <= 0 || last < m->hp.n_ctx, "sequence %d is at position %d: %d %s step(s) would run past n_ctx=%d (thk_model_seq_set / thk_model_reset_kv first)",
            seq, sb.pos_host, n_steps, advance ? "advancing" : "hold-position", m->hp.n_ctx);
    return THK_OK;
}
// One graph of n decode steps (2 <= n <= kMaxGraphSteps), captured on first use.  Consecutive graph launches are ~50 us apart on
// the GPU (measured: single-step replays run 58 us per step slower than 8-step graphs), so a request is served by as few launches
// as possible: floor(n / kMaxGraphSteps) graphs of kMaxGraphSteps steps and one graph of exactly the remainder.
static int ensure_multi_graph(thk_model* m, int seq, int n) {
    thk_ctx* ctx = m->ctx;
    SeqBuf& sb = m->seqs[seq];
    if (sb.multi.count(n)) { sb.multi_used[n] = ++sb.multi_clock; return THK_OK; }
    int rc = THK_OK;
    hipGraph_t g = nullptr; hipGraphExec_t x = nullptr;
    HIPCHK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < n && rc == THK_OK; ++k) rc = enqueue_step(m, seq, nullptr);
    hipError_t e = hipStreamEndCapture(ctx->stream, &g);
    if (rc != THK_OK) { if (g) hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return fail(ctx, THK_ERR_HIP, "hipStreamEndCapture (%d-step graph): %s", n, hipGetErrorString(e));
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { hipGraphDestroy(g); return fail(ctx, THK_ERR_HIP, "hipGraphInstantiate (%d-step graph): %s", n, hipGetErrorString(e)); }
    // a sequence keeps at most kMaxMultiGraphs step counts (n * ~161 nodes each): the least recently replayed one makes room
    if ((int)sb.multi.size() >= kMaxMultiGraphs) {
        auto victim = sb.multi.begin();
        for (auto it = sb.multi.begin(); it != sb.multi.end(); ++it) if (sb.multi_used[it->first] < sb.multi_used[victim->first]) victim = it;
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) {   // a replay of the victim may still be in flight (rare path: a 7th distinct step count)
            hipGraphExecDestroy(x); hipGraphDestroy(g);
            return fail(ctx, THK_ERR_HIP, "hipStreamSynchronize before evicting a %d-step graph failed", victim->first);
        }
        hipGraphExecDestroy(victim->second.second); hipGraphDestroy(victim->second.first);
        sb.multi_used.erase(victim->first); sb.multi.erase(victim);
    }
    sb.multi[n] = {g, x};
    sb.multi_used[n] = ++sb.multi_clock;
    return THK_OK;
}
extern "C" int thk_model_decode_step(thk_model* m, int32_t seq, int advance) {
    if (!m) return THK_ERR_INVALID;
    REQUIRE(m->ctx, m->finalized && seq >= 0 && seq < m->n_seq, "bad sequence %d (or model not finalized)", seq);
    HIPCHK(m->ctx, hipSetDevice(m->ctx->device));
    int rc = check_room(m, seq, 1, advance);
    if (rc != THK_OK) return rc;
    rc = set_advance(m, seq, advance);
    if (rc != THK_OK) return rc;
    rc = run_step(m, seq);
    if (rc == THK_OK && advance) m->seqs[seq].pos_host += 1;
    return rc;
}
// Capture (without running) every multi-step graph thk_model_decode_steps(n_steps) will replay, so that the first
// timed call does not pay for stream capture + hipGraphInstantiate (several ms for 8 x 161 nodes).
extern "C" int thk_model_prepare_steps(thk_model* m, int32_t seq, int32_t n_steps) {
    if (!m) return THK_ERR_INVALID;
    REQUIRE(m->ctx, m->finalized && seq >= 0 && seq < m->n_seq && n_steps >= 0, "bad sequence %d / step count %d (or model not finalized)", seq, n_steps);
    if (!m->use_graph) return THK_OK;
    HIPCHK(m->ctx, hipSetDevice(m->ctx->device));
    if (n_steps >= kMaxGraphSteps) { int rc = ensure_multi_graph(m, seq, kMaxGraphSteps); if (rc != THK_OK) return rc; }
    const int rem = n_steps % kMaxGraphSteps;
    if (rem >= 2) return ensure_multi_graph(m, seq, rem);
    return THK_OK;
}
extern "C" int thk_model_decode_steps(thk_model* m, int32_t seq, int32_t n_steps, int advance) {
    if (!m) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    REQUIRE(ctx, m->finalized && seq >= 0 && seq < m->n_seq && n_steps >= 0, "bad sequence %d / step count %d (or model not finalized)", seq, n_steps);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = check_room(m, seq, n_steps, advance);
    if (rc != THK_OK) return rc;
    rc = set_advance(m, seq, advance);
    if (rc != THK_OK) return rc;
    SeqBuf& sb = m->seqs[seq];
    int left = n_steps;
    if (m->use_graph) {
        while (left >= 2) {                          // 20 = one 20-step graph; 200 = 6 x 32 + 8
            const int n = left >= kMaxGraphSteps ? kMaxGraphSteps : left;
            if ((rc = ensure_multi_graph(m, seq, n)) != THK_OK) return rc;
            HIPCHK(ctx, hipGraphLaunch(sb.multi[n].second, ctx->stream));
            left -= n;
        }
    }
    while (left-- > 0) { rc = run_step(m, seq); if (rc != THK_OK) return rc; }
    if (advance) sb.pos_host += n_steps;
    return THK_OK;
}
extern "C" int thk_model_profile_step(thk_model* m, int32_t seq, int32_t max_entries, char (*names)[48], float* ms, int32_t* n_out) {
    if (!m || !names || !ms || !n_out) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    REQUIRE(ctx, m->finalized && seq >= 0 && seq < m->n_seq, "bad sequence %d (or model not finalized)", seq);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = check_room(m, seq, 1, 0);
    if (rc != THK_OK) return rc;
    const int advance_before = m->seqs[seq].advance_host;     // a hold-position step, like thk_model_step_trace: the host's position mirror stays exact
    if ((rc = set_advance(m, seq, 0)) != THK_OK) return rc;
    StepProf p;
    rc = enqueue_step(m, seq, &p);
    if (rc == THK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, THK_ERR_HIP, "sync failed while profiling");
    if (advance_before >= 0) { const int rc2 = set_advance(m, seq, advance_before); if (rc == THK_OK) rc = rc2; }
    int n = 0;
    if (rc == THK_OK) {
        for (size_t i = 0; i + 1 < p.events.size() && i < p.names.size() && n < max_entries; ++i, ++n) {
            float t = 0.f;
            hipEventElapsedTime(&t, p.events[i], p.events[i + 1]);
            strncpy(names[n], p.names[i].c_str(), 47); names[n][47] = 0;
            ms[n] = t;
        }
    }
    for (auto ev : p.events) hipEventDestroy(ev);
    *n_out = n;
    return rc;
}

// Development aid (libthk_trace.so, tools/step_trace.py): decode steps with every wave of every launch stamping the 100 MHz
// s_memrealtime counter at four points (kernel entry | activation vector staged | first weight batch consumed | done).  With
// graphs on (the default) TWO hold-position steps are captured into one graph and replayed, and the SECOND one's stamps are
// returned (the first absorbs the replay's start-up); with use_graph = 0 it is one eager step.  The traced steps never advance the
// sequence (the advance flag is forced to 0 and restored), so the host's position mirror stays exact.  Returns
// [n_kernels][kTraceBlocks][8 waves][4] u64 (0 = not stamped) and the launch names in thk_model_profile_step order.
extern "C" int thk_model_step_trace(thk_model* m, int32_t seq, unsigned long long* out, int64_t cap_words, int32_t max_names, char (*names)[48],
                                    int32_t* n_kernels, int32_t* blocks_per_kernel) {
    if (!m || !out || !n_kernels || !blocks_per_kernel) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    REQUIRE(ctx, m->finalized && seq >= 0 && seq < m->n_seq, "bad sequence %d (or model not finalized)", seq);
    if (!trace_compiled()) return fail(ctx, THK_ERR_STATE, "this libthk was built without -DTHK_TRACE (build libthk_trace.so: __graft_entry__.build_libthk(trace=True))");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = check_room(m, seq, 1, 0);
    if (rc != THK_OK) return rc;
    const size_t max_k = 6 * (size_t)(m->l1 - m->l0) + 8;
    const size_t words = max_k * kTraceBlocks * kTraceWords;
    if (!m->trace_buf) HIPCHK(ctx, hipMalloc((void**)&m->trace_buf, words * 8));
    HIPCHK(ctx, hipMemsetAsync(m->trace_buf, 0, words * 8, ctx->stream));
    const int advance_before = m->seqs[seq].advance_host;
    if ((rc = set_advance(m, seq, 0)) != THK_OK) return rc;
    StepProf p;
    p.names_only = true;
    if (m->use_graph) {      // the timeline of a step as it is normally run: a replayed graph (two steps, the SECOND one is recorded)
        hipGraph_t g = nullptr; hipGraphExec_t x = nullptr;
        if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            if (advance_before >= 0) set_advance(m, seq, advance_before);
            return fail(ctx, THK_ERR_HIP, "hipStreamBeginCapture (trace) failed");
        }
        rc = enqueue_step(m, seq, nullptr);
        m->trace_on = true;
        if (rc == THK_OK) rc = enqueue_step(m, seq, &p);
        hipError_t e = hipStreamEndCapture(ctx->stream, &g);
        m->trace_on = false;
        if (rc == THK_OK && e != hipSuccess) rc = fail(ctx, THK_ERR_HIP, "hipStreamEndCapture (trace): %s", hipGetErrorString(e));
        if (rc == THK_OK && hipGraphInstantiate(&x, g, nullptr, nullptr, 0) != hipSuccess) rc = fail(ctx, THK_ERR_HIP, "hipGraphInstantiate (trace)");
        if (rc == THK_OK && hipGraphLaunch(x, ctx->stream) != hipSuccess) rc = fail(ctx, THK_ERR_HIP, "hipGraphLaunch (trace)");
        if (rc == THK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, THK_ERR_HIP, "sync failed while tracing");
        if (x) hipGraphExecDestroy(x);
        if (g) hipGraphDestroy(g);
    } else {
        m->trace_on = true;
        rc = enqueue_step(m, seq, &p);
        m->trace_on = false;
        if (rc == THK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, THK_ERR_HIP, "sync failed while tracing");
    }
    for (auto ev : p.events) hipEventDestroy(ev);
    if (advance_before >= 0) { const int rc2 = set_advance(m, seq, advance_before); if (rc == THK_OK) rc = rc2; }
    if (rc != THK_OK) return rc;
    const size_t nk = p.names.size();
    REQUIRE(ctx, nk <= max_k && (int64_t)(nk * kTraceBlocks * kTraceWords) <= cap_words, "step trace needs %zu words", nk * kTraceBlocks * kTraceWords);
    HIPCHK(ctx, hipMemcpy(out, m->trace_buf, nk * kTraceBlocks * kTraceWords * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nk && (int)i < max_names && names; ++i) { strncpy(names[i], p.names[i].c_str(), 47); names[i][47] = 0; }
    *n_kernels = (int)nk; *blocks_per_kernel = kTraceBlocks;
    return THK_OK;
}

// the three doors thk_model.cpp uses (finalize: warm-up and per-sequence graph capture; thk_model_eval)
int step_enqueue(thk_model* m, int seq) { return enqueue_step(m, seq, nullptr); }
int step_run(thk_model* m, int seq) { return run_step(m, seq); }
int step_set_advance(thk_model* m, int seq, int advance) { return set_advance(m, seq, advance); }
