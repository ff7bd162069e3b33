// thk_host_capi.cpp — plain-C surface of the C++ host layer.
//  * capi_*  : the reference's wasm exports (web/main.cpp:72-179) re-created natively: streamed
//              model load (header, then one record at a time) and a "human message" entry point
//              with the onNewToken / onInferenceComplete / onError callbacks.
//  * thh_*   : small hooks the pytest suite uses to exercise the tokenizer, sampler, fp16
//              converters and ggjt parser without a device, and load/eval/do_inference with one.
#include "thk_host.hpp"

#include <string.h>
#include <map>
#include <mutex>

using namespace th;

#define EXPORT extern "C" __attribute__((visibility("default")))

namespace {
std::shared_ptr<LlamaModel>& g_model = *new std::shared_ptr<LlamaModel>();   // one model per process, like the reference (web/main.cpp:21-28)
thk_ctx* g_ctx = nullptr;
std::string g_transcript, g_last_error;
// heap-allocated and never destroyed: models must not be torn down at static-destruction time,
// when the caller's thk_ctx is already gone
std::map<int64_t, std::shared_ptr<LlamaModel>>& g_handles = *new std::map<int64_t, std::shared_ptr<LlamaModel>>();
int64_t g_next_handle = 1;
}  // namespace

// ---------------------------------------------------------------- capi_* (web/main.cpp:72-179)
EXPORT const char* capi_test_capi() { return "thk host capi"; }
EXPORT int capi_model_begin_load(thk_ctx* ctx) {
    g_ctx = ctx; g_model = std::make_shared<LlamaModel>(); g_transcript.clear(); g_last_error.clear();
    g_model->onError = [](std::string e) { g_last_error = e; };
    g_model->onNewToken = [](std::string, std::string so_far) { g_transcript = so_far; };
    g_model->onInferenceComplete = [](std::string full) { g_transcript = full; };
    return 1;
}
EXPORT int capi_load_model_header(const void* data, double size) {
    return g_model && load_header(g_model.get(), data, (int64_t)size) ? 1 : 0;
}
EXPORT int capi_load_model_weights(const void* data, double fileOffset, double size) {
    return g_model && load_weights(g_model.get(), g_ctx, data, (int64_t)size, 1, (int64_t)fileOffset) ? 1 : 0;
}
EXPORT int capi_model_end_load() { return g_model && post_load_init_model(g_ctx, g_model) ? 1 : 0; }
EXPORT const char* capi_on_human_message(const char* message) {
    if (!g_model || !g_model->dev) return "";
    if (!strncmp(message, "[cmd] reset", 11)) { reset_context(g_model); g_transcript = "context reset"; return g_transcript.c_str(); }
    do_inference(g_ctx, g_model, message);
    return g_transcript.c_str();
}
EXPORT const char* capi_last_error() { return g_last_error.c_str(); }
EXPORT void capi_model_unload() { g_model.reset(); }
EXPORT void capi_set_prompt_prefill(int on) { if (g_model) g_model->prefillPrompt = on != 0; }
EXPORT void capi_set_sampler(int top_k, float top_p, float temp, float repeat_penalty) {
    if (g_model) g_model->sampler = SamplerParams{top_k, top_p, temp, repeat_penalty, false};
}

// ---------------------------------------------------------------- thh_* test hooks (no device)
EXPORT void thh_fp16_to_fp32(const uint16_t* h, float* out, int64_t n) { for (int64_t i = 0; i < n; ++i) out[i] = ggml_compute_fp16_to_fp32(h[i]); }
EXPORT void thh_fp32_to_fp16(const float* f, uint16_t* out, int64_t n) { for (int64_t i = 0; i < n; ++i) out[i] = ggml_compute_fp32_to_fp16(f[i]); }

static LlamaVocab make_vocab(const char* blob, const int32_t* lens, const float* scores, int n) {
    LlamaVocab v; v.id_to_token.resize(n);
    int64_t off = 0;
    for (int i = 0; i < n; ++i) { std::string w(blob + off, lens[i]); off += lens[i]; v.token_to_id[w] = i; v.id_to_token[i] = {w, scores[i]}; }
    return v;
}
EXPORT int thh_tokenize(const char* blob, const int32_t* lens, const float* scores, int n_vocab, const char* text, int text_len, int add_bos,
                        int32_t* out, int cap) {
    const LlamaVocab v = make_vocab(blob, lens, scores, n_vocab);
    const auto ids = tk_llama_tokenize(v, std::string(text, text_len), add_bos != 0);
    for (size_t i = 0; i < ids.size() && (int)i < cap; ++i) out[i] = ids[i];
    return (int)ids.size();
}
EXPORT void thh_sample(uint32_t seed, const float* logits, int n_vocab, int top_k, float top_p, float temp, float repeat_penalty,
                       const int32_t* last_n, int n_last, int n_draws, int32_t* out) {
    std::mt19937 rng(seed);
    std::vector<float> lg(logits, logits + n_vocab);
    std::vector<tk_llama_token> last(last_n, last_n + n_last);
    for (int i = 0; i < n_draws; ++i) out[i] = llama_sample_top_p_top_k(rng, n_vocab, last, top_k, top_p, temp, repeat_penalty, lg);
}
EXPORT int thh_parse_header(const void* data, int64_t size, int32_t* hp7, int64_t* consumed, int32_t* n_tokens_seen) {
    LlamaModel m; m.onError = [](std::string e) { g_last_error = e; };
    if (!load_header(&m, data, size, consumed)) return 0;
    const int32_t v[7] = {m.n_vocab, m.n_embd, m.n_mult, m.n_head, m.n_layer, m.n_rot, m.f16};
    memcpy(hp7, v, sizeof v);
    *n_tokens_seen = (int32_t)m.vocab.id_to_token.size();
    return 1;
}
EXPORT int thh_parse_tensor(const void* data, int64_t size, int64_t file_off, char* name, int name_cap, int32_t* type, int64_t* shape4,
                            int64_t* ne01, int64_t* data_off, int64_t* data_bytes, int64_t* record_bytes) {
    GgjtTensorInfo ti; std::string err;
    if (!parse_tensor_record(data, size, file_off, &ti, &err)) { g_last_error = err; return 0; }
    strncpy(name, ti.name.c_str(), name_cap - 1); name[name_cap - 1] = 0;
    *type = ti.type == TensorType_F16 ? 1 : 0;
    shape4[0] = ti.shape.l; shape4[1] = ti.shape.b; shape4[2] = ti.shape.r; shape4[3] = ti.shape.c;
    ne01[0] = ti.ne0; ne01[1] = ti.ne1;
    *data_off = ti.data_offset; *data_bytes = ti.data_bytes; *record_bytes = ti.record_bytes;
    return 1;
}
EXPORT int thh_tensor_shape_roundtrip(int64_t l, int64_t b, int64_t r, int64_t c, int64_t* out5) {
    TensorShape s{l, b, r, c};
    out5[4] = s.get_total_num_elements();
    s.canonicalize();
    out5[0] = s.l; out5[1] = s.b; out5[2] = s.r; out5[3] = s.c;
    return 1;
}

// ---------------------------------------------------------------- thh_* with a device
EXPORT int64_t thh_load_file(thk_ctx* ctx, const char* path, int lmhead_mode) {
    auto m = load_llama_file(ctx, path, lmhead_mode);
    if (!m) return 0;
    m->onError = [](std::string e) { g_last_error = e; };
    g_handles[g_next_handle] = m;
    return g_next_handle++;
}
EXPORT void thh_free(int64_t h) { g_handles.erase(h); }
EXPORT int thh_hparams(int64_t h, int32_t* hp7) {
    auto it = g_handles.find(h); if (it == g_handles.end()) return 0;
    auto& m = *it->second;
    const int32_t v[7] = {m.n_vocab, m.n_embd, m.n_mult, m.n_head, m.n_layer, m.n_rot, m.f16};
    memcpy(hp7, v, sizeof v);
    return 1;
}
EXPORT int thh_set_sampler(int64_t h, int top_k, float top_p, float temp, float repeat_penalty) {
    auto it = g_handles.find(h); if (it == g_handles.end()) return 0;
    it->second->sampler = SamplerParams{top_k, top_p, temp, repeat_penalty, false};
    return 1;
}
EXPORT int thh_set_prefill(int64_t h, int on) {
    auto it = g_handles.find(h); if (it == g_handles.end()) return 0;
    it->second->prefillPrompt = on != 0;
    return 1;
}
EXPORT int thh_eval(int64_t h, const int32_t* tokens, int n, int n_past, float* logits_out) {
    auto it = g_handles.find(h); if (it == g_handles.end()) return -1;
    auto m = it->second;
    std::vector<tk_llama_token> t(tokens, tokens + n);
    const tk_llama_token tok = th_eval(m->ctx, m, t.data(), n, n_past);
    if (logits_out) memcpy(logits_out, m->logits.data(), m->logits.size() * 4);
    return tok;
}
EXPORT int thh_do_inference(int64_t h, const char* prompt, int32_t* n_past_out, char* text_out, int text_cap) {
    auto it = g_handles.find(h); if (it == g_handles.end()) return 0;
    auto m = it->second;
    std::string full; int n_tok = 0;
    m->onNewToken = [&](std::string, std::string) { ++n_tok; };
    m->onInferenceComplete = [&](std::string f) { full = f; };
    do_inference(m->ctx, m, prompt);
    if (n_past_out) *n_past_out = m->n_past;
    if (text_out && text_cap > 0) { strncpy(text_out, full.c_str(), text_cap - 1); text_out[text_cap - 1] = 0; }
    return n_tok;
}
EXPORT void thh_reset(int64_t h) { auto it = g_handles.find(h); if (it != g_handles.end()) reset_context(it->second); }
EXPORT const char* thh_last_error() { return g_last_error.c_str(); }
