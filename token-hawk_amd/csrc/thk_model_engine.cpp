// thk_model_engine.cpp — host side of the optional persistent decode engine (thk_engine.hip, tunable engine=1): eligibility and
// LDS plan, the per-sequence op program, the development timeline read-back and the device error word.
#include "thk_internal.hpp"

// ----- persistent engine (thk_engine.hip): eligibility, LDS plan and the per-sequence op program
bool engine_plan(thk_model* m) {
    thk_ctx* ctx = m->ctx;
    const int E = m->hp.n_embd, H = m->hp.n_head, D = E / H, F = m->n_ff, V = m->hp.n_vocab, T = m->hp.n_ctx;
    if (tun(ctx, "engine") == 0) return false;
    if (D != 64 && D != 128) return false;
    if (E % 512 != 0 || F % 256 != 0 || E > 6144 || (V & 1)) return false;            // row pairs, one-sweep norm gather, whole pieces
    // a consumer wave keeps the epilogue operands of ALL its units of an op one per lane (rdlane(rv, i), thk_engine_body.inc):
    // at most 64 units per wave, i.e. 192 per CU, for the ops that carry operands (qkv: 3E/2 units, wo / w2: E/2)
    if ((3 * E / 2 + ctx->n_cu - 1) / ctx->n_cu > 192) return false;
    if ((m->flags & THK_STAGE_HEAD) && m->lm_mode != THK_LMHEAD_CORRECT) return false;  // the Q1-faithful combine stays on the launch path
    if (m->skip_kernel || m->kv_f16 || H > ctx->n_cu) return false;   // the engine reads the reference's f32 cache
    int S = 1;
    while (S * 2 <= kMaxSplit && S * 2 * H <= ctx->n_cu) S *= 2;
    const int v1 = ((E + 511) / 512) * 2048, v0 = ((std::max(E, F) + 511) / 512) * 2048;
    const long budget = 160 * 1024 - (long)engine_lds_bytes(0, v0, v1);
    const int NS = (int)(budget / kEngSlotBytes);
    if (NS < 3) return false;
    m->eng_NS = NS > 8 ? 8 : NS; m->eng_v0 = v0; m->eng_v1 = v1; m->eng_nsplit = S; m->eng_tc = (T + S - 1) / S;
    return true;
}
static void eng_unit_geometry(EngOp& o) {       // a unit = two rows of C f16 = 4C bytes = C/256 pieces of 1 KiB
    o.row_bytes = o.C * 2;
    const int pieces = o.C / 256;
    o.fpu = (pieces + 15) / 16;
    o.pieces_last = pieces - 16 * (o.fpu - 1);
}
int engine_build_program(thk_model* m, SeqBuf& sb) {
    thk_ctx* ctx = m->ctx;
    const int E = m->hp.n_embd, H = m->hp.n_head, D = E / H, F = m->n_ff, V = m->hp.n_vocab, T = m->hp.n_ctx, S = m->eng_nsplit;
    const int nl = m->l1 - m->l0;
    unsigned long long* XG0 = m->eng_gran; unsigned long long* XG1 = XG0 + E; unsigned long long* QG = XG1 + E;
    unsigned long long* OG = QG + 3 * (size_t)E; unsigned long long* UG = OG + E; unsigned long long* PG = UG + F;
    (void)H; (void)S; (void)D;
    const float* xin = (m->flags & THK_STAGE_EMBED) ? m->x : sb.hidden_in;
    std::vector<EngOp> ops;
    int prev_w2 = -1;
    for (int i = 0; i < nl; ++i) {
        const LayerW& L = m->layers[i];
        float* kc = kcache_of(m, sb, i);
        float* vc = vcache_of(m, sb, i);
        EngOp q{};    // rms_norm*gain -> wq,wk,wv -> RoPE -> K/V append (th-llama.cpp:299-339)
        q.kind = EOP_QKV; q.n_units = 3 * E / 2; q.C = E; eng_unit_geometry(q);
        q.W[0] = L.wq; q.W[1] = L.wk; q.W[2] = L.wv; q.gain = L.attention_norm;
        q.in_src = i == 0 ? EIN_PLAIN : EIN_GRAN; q.in_ptr = i == 0 ? (const void*)xin : (const void*)XG0; q.in_n = E; q.in_tag_op = prev_w2; q.in_dst = 1;
        q.out_g = QG; q.kcache = kc; q.vcache = vc;
        const int iq = (int)ops.size(); ops.push_back(q);
        EngOp at{};   // attention over the cache in place (th-llama.cpp:341-397)
        at.kind = EOP_ATTN; at.in_tag_op = iq; at.qg = QG; at.pg = PG; at.out_g = OG; at.kcache = kc; at.vcache = vc;
        const int ia = (int)ops.size(); ops.push_back(at);
        EngOp o{};    // wo -> + residual (th-llama.cpp:401-413)
        o.kind = EOP_WO; o.n_units = E / 2; o.C = E; eng_unit_geometry(o); o.W[0] = L.wo;
        o.in_src = EIN_GRAN; o.in_ptr = OG; o.in_n = E; o.in_tag_op = ia; o.in_dst = 0;
        o.resid_src = i == 0 ? 2 : 1; o.resid_ptr = i == 0 ? (const void*)xin : (const void*)XG0; o.out_g = XG1;
        const int io = (int)ops.size(); ops.push_back(o);
        EngOp g{};    // rms_norm*gain -> w1,w3 -> silu*gate (th-llama.cpp:415-438)
        g.kind = EOP_W13; g.n_units = F; g.C = E; eng_unit_geometry(g); g.dual = (g.row_bytes % 4096 == 0) ? 1 : 2; g.W[0] = L.w1; g.W[1] = L.w3; g.gain = L.ffn_norm;
        g.in_src = EIN_GRAN; g.in_ptr = XG1; g.in_n = E; g.in_tag_op = io; g.in_dst = 1; g.out_g = UG;
        const int ig = (int)ops.size(); ops.push_back(g);
        EngOp d{};    // w2 -> + residual (th-llama.cpp:440-451)
        d.kind = EOP_W2; d.n_units = E / 2; d.C = F; eng_unit_geometry(d); d.W[0] = L.w2;
        d.in_src = EIN_GRAN; d.in_ptr = UG; d.in_n = F; d.in_tag_op = ig; d.in_dst = 0;
        d.resid_src = 1; d.resid_ptr = XG1; d.out_g = XG0;
        if (i == nl - 1) d.out_plain = (m->flags & THK_STAGE_HEAD) ? m->x : sb.hidden_out;
        prev_w2 = (int)ops.size(); ops.push_back(d);
    }
    if (m->flags & THK_STAGE_HEAD) {   // final norm -> lm-head -> greedy keys (th-llama.cpp:240-268, :826-838)
        EngOp h{};
        h.kind = EOP_HEAD; h.n_units = V / 2; h.C = E; eng_unit_geometry(h); h.W[0] = m->output; h.gain = m->norm;
        h.in_src = EIN_GRAN; h.in_ptr = XG0; h.in_n = E; h.in_tag_op = prev_w2; h.in_dst = 1; h.out_plain = sb.logits;
        ops.push_back(h);
    }
    REQUIRE(ctx, ops.size() < 511, "engine program of %zu ops does not fit the 9-bit op tag", ops.size());
    HIPCHK(ctx, hipMalloc((void**)&sb.eng_ops, ops.size() * sizeof(EngOp)));
    HIPCHK(ctx, hipMemcpy(sb.eng_ops, ops.data(), ops.size() * sizeof(EngOp), hipMemcpyHostToDevice));
    sb.eng_n_ops = (int)ops.size();
    return THK_OK;
}

// Development aid: copies the engine timeline of the last step ([n_cu][n_ops][8] u64, see thk_engine.hip) to the host.
extern "C" int thk_model_engine_trace(thk_model* m, unsigned long long* out, int64_t cap_words, int32_t* n_cu, int32_t* n_ops) {
    if (!m || !out) return THK_ERR_INVALID;
    thk_ctx* ctx = m->ctx;
    REQUIRE(ctx, m->finalized && m->engine && m->eng_trace, "no engine timeline (set the tunable engine_trace=1 before finalize)");
    const int64_t words = (int64_t)ctx->n_cu * m->seqs[0].eng_n_ops * 8;
    REQUIRE(ctx, cap_words >= words, "engine timeline needs %lld words", (long long)words);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(out, m->eng_trace, (size_t)words * 8, hipMemcpyDeviceToHost));
    if (n_cu) *n_cu = ctx->n_cu;
    if (n_ops) *n_ops = m->seqs[0].eng_n_ops;
    return THK_OK;
}
// Bounded in-launch waits raise a device-side error word instead of hanging; surface it.
int check_engine_error(thk_model* m) {
    if (m->engine && m->eng_words) {
        unsigned e = 0;
        HIPCHK(m->ctx, hipMemcpyAsync(&e, m->eng_words + 32, 4, hipMemcpyDeviceToHost, m->ctx->stream));
        HIPCHK(m->ctx, hipStreamSynchronize(m->ctx->stream));
        if (e) {
            HIPCHK(m->ctx, hipMemsetAsync(m->eng_words + 32, 0, 4, m->ctx->stream));
            return fail(m->ctx, THK_ERR_STATE, "decode engine: bounded wait timed out (code %u, op %u, workgroup %u)", (e >> 24) & 0x7f, (e >> 12) & 0xfff, e & 0xfff);
        }
    }
    return THK_OK;
}

