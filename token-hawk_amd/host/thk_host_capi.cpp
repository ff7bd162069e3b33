// thk_host_capi.cpp — plain-C surface of the C++ host layer.
//  * capi_*  : the reference's wasm exports (web/main.cpp:72-179) re-created natively: streamed
//              model load (header, then one record at a time) and a "human message" entry point
//              with the onNewToken / onInferenceComplete / onError callbacks.
//  * thh_*   : small hooks the pytest suite uses to exercise the tokenizer, sampler, fp16
//              converters and ggjt parser without a device, and load/eval/do_inference with one.
#include "thk_host.hpp"

#include <string.h>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>

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
// Same names, argument lists and return types as the reference's wasm exports:
//   void capi_model_begin_load()                                          web/main.cpp:83-86
//   void capi_load_model_header(void* data, double dataSize)              :94-97
//   void capi_load_model_weights(void* data, double offset, double size)  :100-104
//   bool capi_model_end_load()                                            :138-157
//   void capi_on_human_message(const char* str)                           :160-179
// The reference gets its device from emscripten at start-up (`__main__`, :75-78); natively the embedder hands one over
// with capi_set_context(thk_ctx*) before capi_model_begin_load.  The two JS hooks the reference calls
// (updateMessageText(id, text), :106-110; sendChatMessage via sendNewBotMessage(text, id), :116-120) are C callbacks
// registered with capi_set_ui_hooks.  Like the browser build (th-llama.cpp:176-198, :733-799) capi_on_human_message does
// not block: inference runs on a worker thread, text streams through the hooks, and a message that arrives while a
// reply is still being generated is ignored (web/main.cpp:172).  capi_wait_idle() joins the worker (tests, CLI-style hosts).
namespace {
typedef void (*update_message_text_fn)(const char* message_id, const char* text);
typedef void (*send_new_bot_message_fn)(const char* text, const char* message_id);
update_message_text_fn g_update_text = nullptr;
send_new_bot_message_fn g_send_bot = nullptr;
std::thread g_worker;
std::atomic<bool> g_inference_complete{true};
std::mutex g_text_mutex;                   // guards g_transcript / g_last_error / g_last_bot_id against the worker
std::string g_last_bot_id;
int g_bot_seq = 0;
void join_worker() { if (g_worker.joinable()) g_worker.join(); }
void new_bot_message(const char* prefix, const std::string& text) {
    std::string id;
    { std::lock_guard<std::mutex> l(g_text_mutex); g_bot_seq += 1; id = std::string(prefix) + std::to_string(g_bot_seq); g_last_bot_id = id; }
    if (g_send_bot) g_send_bot(text.c_str(), id.c_str());
}
}  // namespace

EXPORT const char* capi_test_capi() { return "thk host capi"; }
EXPORT void capi_set_context(thk_ctx* ctx) { g_ctx = ctx; }
EXPORT void capi_set_ui_hooks(update_message_text_fn update_text, send_new_bot_message_fn send_bot) { g_update_text = update_text; g_send_bot = send_bot; }
EXPORT void capi_model_begin_load() {
    join_worker();
    g_model = std::make_shared<LlamaModel>();
    { std::lock_guard<std::mutex> l(g_text_mutex); g_transcript.clear(); g_last_error.clear(); }
    g_model->onError = [](std::string e) { std::lock_guard<std::mutex> l(g_text_mutex); g_last_error = e; };   // load-time errors; replaced in end_load
}
EXPORT void capi_load_model_header(void* data, double dataSizeD) {
    if (g_model) load_header(g_model.get(), data, (int64_t)dataSizeD);
}
EXPORT void capi_load_model_weights(void* data, double weightsBeginOffsetD, double dataSizeD) {
    if (g_model && !g_model->loadFailed) load_weights(g_model.get(), g_ctx, data, (int64_t)dataSizeD, 1, (int64_t)weightsBeginOffsetD);
}
EXPORT bool capi_model_end_load() {
    if (!g_model || g_model->loadFailed || !post_load_init_model(g_ctx, g_model)) { if (g_model) g_model->loadFailed = true; return false; }
    g_model->onNewToken = [](std::string, std::string message) {            // web/main.cpp:112-114
        std::string id;
        { std::lock_guard<std::mutex> l(g_text_mutex); g_transcript = message; id = g_last_bot_id; }
        if (g_update_text) g_update_text(id.c_str(), message.c_str());
    };
    g_model->onInferenceComplete = [](std::string message) {                // :122-128
        { std::lock_guard<std::mutex> l(g_text_mutex); g_transcript = message; }
        g_inference_complete = true;
    };
    g_model->onError = [](std::string message) {                            // :130-135
        { std::lock_guard<std::mutex> l(g_text_mutex); g_last_error = message; }
        new_bot_message("bot-error-msg-", message);
        g_inference_complete = true;
    };
    g_inference_complete = true;
    return true;
}
EXPORT void capi_on_human_message(const char* str) {
    if (!g_model || !g_model->dev || !str) return;
    const std::string input(str);
    if (input == "[cmd] reset") {                                           // :164-170
        join_worker();
        reset_context(g_model);
        new_bot_message("bot-msg-", "LLM context reset.");
        return;
    }
    if (!g_inference_complete.load()) return;                               // a reply is still being generated: ignored, as in the reference
    join_worker();
    g_inference_complete = false;
    new_bot_message("bot-msg-", "--");
    g_worker = std::thread([input] {
        do_inference(g_ctx, g_model, input);
        g_inference_complete = true;                                        // also on the early-return paths that reported through onError
    });
}
// ---- native conveniences beyond the reference's exports (tests and CLI-style embedders)
EXPORT int capi_inference_complete() { return g_inference_complete.load() ? 1 : 0; }
EXPORT void capi_wait_idle() { join_worker(); }
EXPORT const char* capi_transcript() { static std::string copy; std::lock_guard<std::mutex> l(g_text_mutex); copy = g_transcript; return copy.c_str(); }
EXPORT const char* capi_last_error() { static std::string copy; std::lock_guard<std::mutex> l(g_text_mutex); copy = g_last_error; return copy.c_str(); }
EXPORT void capi_model_unload() { join_worker(); g_model.reset(); }
EXPORT void capi_set_prompt_prefill(int on) { if (g_model) g_model->prefillPrompt = on != 0; }
EXPORT void capi_set_greedy_device_loop(int on) { if (g_model) g_model->greedyDeviceLoop = on != 0; }
EXPORT void capi_set_device_topk(int on) { if (g_model) g_model->deviceTopK = on != 0; }
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
// The candidate path of the sampler with the device's part played by the CPU: the K largest raw logits, value descending and ties by
// ascending id (exactly what thk_topk_f32 returns), feed llama_sample_from_topk; a refusal (tie / too few candidates) takes the full
// path, as th_eval does.  *n_fast counts the draws that did not need the whole vector.
EXPORT void thh_sample_topk(uint32_t seed, const float* logits, int n_vocab, int top_k, float top_p, float temp, float repeat_penalty,
                            const int32_t* last_n, int n_last, int n_draws, int32_t* out, int32_t* n_fast) {
    std::mt19937 rng(seed);
    std::vector<float> lg(logits, logits + n_vocab);
    std::vector<tk_llama_token> last(last_n, last_n + n_last);
    *n_fast = 0;
    for (int i = 0; i < n_draws; ++i) {
        const int K = llama_topk_candidates_needed(n_vocab, last, top_k, repeat_penalty);
        std::vector<int32_t> order((size_t)n_vocab);
        for (int j = 0; j < n_vocab; ++j) order[j] = j;
        std::partial_sort(order.begin(), order.begin() + K, order.end(), [&](int32_t a, int32_t b) { return lg[a] > lg[b] || (lg[a] == lg[b] && a < b); });
        std::vector<float> cv((size_t)K); std::vector<int32_t> ci(order.begin(), order.begin() + K);
        for (int j = 0; j < K; ++j) cv[j] = lg[ci[j]];
        tk_llama_token tok = -1;
        if (temp > 0 && llama_sample_from_topk(rng, n_vocab, last, top_k, top_p, temp, repeat_penalty, cv.data(), ci.data(), K, &tok)) { out[i] = tok; *n_fast += 1; }
        else out[i] = llama_sample_top_p_top_k(rng, n_vocab, last, top_k, top_p, temp, repeat_penalty, lg);
    }
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
// A LlamaModel without a file: the device model is filled with the seeded synthetic weights (thk_model_fill_synthetic) and the
// vocabulary is the caller's (tests/ggjt.py toy_vocab).  bench.py times th::do_inference on the synthetic 7B through this (extras.host_api).
EXPORT int64_t thh_make_synthetic(thk_ctx* ctx, const int32_t* hp6 /* n_vocab n_embd n_mult n_head n_layer n_ctx */, const char* blob, const int32_t* lens,
                                  const float* scores, uint64_t seed, float sigma) {
    auto m = std::make_shared<LlamaModel>();
    m->onError = [](std::string e) { g_last_error = e; };
    m->n_vocab = hp6[0]; m->n_embd = hp6[1]; m->n_mult = hp6[2]; m->n_head = hp6[3]; m->n_layer = hp6[4]; m->n_ctx = hp6[5];
    m->n_rot = m->n_embd / m->n_head;
    m->vocab = make_vocab(blob, lens, scores, m->n_vocab);
    m->ctx = ctx;
    m->rng = std::mt19937(780658349);   // th-llama-loader.cpp:332-333
    thk_hparams hp{m->n_vocab, m->n_embd, m->n_mult, m->n_head, m->n_layer, m->n_ctx};
    if (thk_model_create(ctx, &hp, 0, m->n_layer, THK_STAGE_EMBED | THK_STAGE_HEAD, 1, &m->dev) != THK_OK || thk_model_fill_synthetic(m->dev, seed, sigma) != THK_OK ||
        thk_model_finalize(m->dev) != THK_OK) {
        g_last_error = std::string("thh_make_synthetic: ") + thk_last_error(ctx);
        return 0;
    }
    g_handles[g_next_handle] = m;
    return g_next_handle++;
}
EXPORT int thh_set_step_limit(int64_t h, int64_t steps) {            // prompt tokens + generated tokens per do_inference call (<= the reference's 500)
    auto it = g_handles.find(h); if (it == g_handles.end()) return 0;
    it->second->stepLimit = steps;
    return 1;
}
EXPORT int thh_collect_stats(int64_t h, int on) {                    // clears the counters
    auto it = g_handles.find(h); if (it == g_handles.end()) return 0;
    it->second->collectStats = on != 0; it->second->stats.clear();
    return 1;
}
// out8: eval_s topk_s readback_s draw_s n_eval n_topk n_readback t_begin; step_end[cap]: wall-clock seconds at the end of every step; returns the number of steps
EXPORT int thh_stats(int64_t h, double* out8, double* step_end, int cap) {
    auto it = g_handles.find(h); if (it == g_handles.end()) return -1;
    const LlamaModel::LoopStats& s = it->second->stats;
    const double v[8] = {s.eval_s, s.topk_s, s.readback_s, s.draw_s, (double)s.n_eval, (double)s.n_topk, (double)s.n_readback, s.t_begin};
    memcpy(out8, v, sizeof v);
    for (size_t i = 0; i < s.step_end.size() && (int)i < cap; ++i) step_end[i] = s.step_end[i];
    return (int)s.step_end.size();
}
EXPORT double thh_now() { return get_time_seconds(); }
EXPORT thk_model* thh_device_model(int64_t h) { auto it = g_handles.find(h); return it == g_handles.end() ? nullptr : it->second->dev; }   // for thk_model_get_tensor & co.
EXPORT thk_model* capi_device_model() { return g_model ? g_model->dev : nullptr; }
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
EXPORT int thh_set_greedy_device_loop(int64_t h, int on) {
    auto it = g_handles.find(h); if (it == g_handles.end()) return 0;
    it->second->greedyDeviceLoop = on != 0;
    return 1;
}
EXPORT int thh_set_device_topk(int64_t h, int on) {
    auto it = g_handles.find(h); if (it == g_handles.end()) return 0;
    it->second->deviceTopK = on != 0;
    return 1;
}
// TensorBuffer (th.hpp:83-148) on a real device: allocate + upload through the constructor, move-construct and move-assign
// (the source must be left empty, the destination must own the SAME device allocation), relabel the shape and restore it,
// download, free.  Returns a bit mask of the checks that passed (0x3F = all).
EXPORT int thh_tensor_buffer_semantics(thk_ctx* ctx, const float* data, int64_t rows, int64_t cols, float* back) {
    int ok = 0;
    TensorBuffer a(data, TensorShape{0, 0, rows, cols}, TensorType_F32, /*backup=*/true, ctx);
    if (a.is_valid() && a.gpu && a.get_size_bytes() == (size_t)(rows * cols * 4) && a.cpuBackup.size() == a.get_size_bytes()) ok |= 1;
    void* dev = a.device_ptr();
    TensorBuffer b(std::move(a));                                 // move construction
    if (!a.gpu && !a.is_valid() && b.device_ptr() == dev && b.is_valid()) ok |= 2;
    TensorBuffer c;
    c = std::move(b);                                             // move assignment
    if (!b.gpu && c.device_ptr() == dev) ok |= 4;
    c.shape = TensorShape{0, 0, 1, rows * cols};                  // relabel without moving data (th-llama.cpp:343-361), then restore
    const bool relabel_ok = c.get_size_bytes() == (size_t)(rows * cols * 4);
    c.reset_shape();
    if (relabel_ok && c.shape == TensorShape{0, 0, rows, cols}) ok |= 8;
    if (c.download(back)) ok |= 16;
    c.free_buffers();
    if (!c.gpu && !c.device_ptr()) ok |= 32;
    return ok;
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
