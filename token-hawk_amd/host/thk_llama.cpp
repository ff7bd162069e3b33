// thk_llama.cpp — tensors, model state, token loop, tokenizer and sampler of the host layer.
// Behavioural mirror of th.cpp:150-229 / :294-359 and th-llama.cpp:111-238, :464-727, :802-1108,
// written against the libthk C-ABI.  See thk_host.hpp for what is intentionally different.
#include "thk_host.hpp"

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <limits>
#include <numeric>
#include <queue>

namespace th {

// ------------------------------------------------------------------ tensor types
std::string get_TensorType_name(TensorType dt) {
    switch (dt) { case TensorType_F16: return "f16"; case TensorType_F32: return "f32"; default: return "unknown"; }
}
size_t get_TensorType_size(TensorType dt) { return dt == TensorType_F16 ? 2 : dt == TensorType_F32 ? 4 : 0; }

int64_t TensorShape::get_total_num_elements() const {
    if (l == 0 && b == 0 && r == 0 && c == 0) return 0;
    int64_t n = 1;
    for (int64_t d : {l, b, r, c}) if (d > 0) n *= d;
    return n;
}
std::string TensorShape::to_string() const {
    return "L:" + std::to_string(l) + " B:" + std::to_string(b) + " R:" + std::to_string(r) + " C:" + std::to_string(c);
}
void TensorShape::canonicalize() {
    if (l == 1) l = 0;
    if (b == 1) b = 0;
    if (r == 0) r = 1;
}

TensorBuffer::TensorBuffer(TensorShape s, TensorType t, thk_ctx* c) : shape(s), originalShape(s), type(t), ctx(c) {
    if (ctx && get_size_bytes() > 0 && thk_buf_alloc(ctx, get_size_bytes(), &gpu) != THK_OK) gpu = nullptr;
}
TensorBuffer::TensorBuffer(const void* data, TensorShape s, TensorType t, bool backup, thk_ctx* c) : TensorBuffer(s, t, c) {
    if (backup && data) { cpuBackup.resize(get_size_bytes()); memcpy(cpuBackup.data(), data, cpuBackup.size()); }
    if (gpu && data) upload_data_to_gpu(data);
}
TensorBuffer& TensorBuffer::operator=(TensorBuffer&& o) noexcept {
    if (this != &o) {
        free_buffers();
        shape = o.shape; originalShape = o.originalShape; type = o.type; ctx = o.ctx; gpu = o.gpu; cpuBackup = std::move(o.cpuBackup);
        o.gpu = nullptr; o.shape = {}; o.originalShape = {}; o.type = TensorType_Unknown;
    }
    return *this;
}
size_t TensorBuffer::get_size_bytes() const { return (size_t)shape.get_total_num_elements() * get_TensorType_size(type); }
bool TensorBuffer::upload_data_to_gpu(const void* data) { return gpu && thk_buf_upload(ctx, gpu, 0, data, get_size_bytes()) == THK_OK; }
bool TensorBuffer::download(void* out) const { return gpu && thk_buf_download(ctx, gpu, 0, out, get_size_bytes()) == THK_OK; }
void TensorBuffer::free_buffers() {
    if (gpu) { thk_buf_free(ctx, gpu); gpu = nullptr; }
}

// ------------------------------------------------------------------ fp16 <-> fp32 (th.cpp:312-359)
// IEEE binary16 <-> binary32 via the exponent-rebias / magic-bias construction GGML uses; exact
// for every finite value, RNE on narrowing, NaN -> 0x7E00.
static inline float from_bits(uint32_t w) { float f; memcpy(&f, &w, 4); return f; }
static inline uint32_t to_bits(float f) { uint32_t w; memcpy(&w, &f, 4); return w; }

float ggml_compute_fp16_to_fp32(ggml_fp16_t h) {
    const uint32_t w = (uint32_t)h << 16, sign = w & 0x80000000u, two_w = w + w;
    const float norm = from_bits((two_w >> 4) + (0xE0u << 23)) * 0x1.0p-112f;
    const float denorm = from_bits((two_w >> 17) | (126u << 23)) - 0.5f;
    return from_bits(sign | (two_w < (1u << 27) ? to_bits(denorm) : to_bits(norm)));
}
ggml_fp16_t ggml_compute_fp32_to_fp16(float f) {
    float base = (fabsf(f) * 0x1.0p+112f) * 0x1.0p-110f;
    const uint32_t w = to_bits(f), shl1 = w + w, sign = w & 0x80000000u;
    uint32_t bias = shl1 & 0xFF000000u;
    if (bias < 0x71000000u) bias = 0x71000000u;
    base = from_bits((bias >> 1) + 0x07800000u) + base;
    const uint32_t bits = to_bits(base);
    const uint32_t nonsign = ((bits >> 13) & 0x00007C00u) + (bits & 0x00000FFFu);
    return (ggml_fp16_t)((sign >> 16) | (shl1 > 0xFF000000u ? 0x7E00u : nonsign));
}

// ------------------------------------------------------------------ timing (th.cpp:23-87)
double get_time_seconds() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
std::string descriptive_stats(std::vector<double> d, const std::string& unit) {
    if (d.empty()) return "no samples";
    std::sort(d.begin(), d.end());
    const double n = (double)d.size(), mean = std::accumulate(d.begin(), d.end(), 0.0) / n;
    double var = 0; for (double x : d) var += (x - mean) * (x - mean);
    auto pct = [&](double p) { return d[(size_t)std::min(n - 1, std::max(0.0, p * (n - 1)))]; };
    char buf[256];
    snprintf(buf, sizeof buf, "n=%zu mean=%.3f median=%.3f stddev=%.3f p99=%.3f p95=%.3f p5=%.3f p1=%.3f%s", d.size(), mean, pct(0.5),
             sqrt(var / n), pct(0.99), pct(0.95), pct(0.05), pct(0.01), unit.c_str());
    return buf;
}

// ------------------------------------------------------------------ model
LlamaModel::~LlamaModel() {
    if (dev) thk_model_destroy(dev);
}
void build_pipelines_llama(thk_ctx*, std::shared_ptr<LlamaModel>) {}   // reference: runtime WGSL compilation (th-llama.cpp:66-76)

static void report_error(LlamaModel& m, const std::string& msg) {
    fprintf(stderr, "Warning: %s\n", msg.c_str());
    if (m.onError) m.onError(msg);
}

void reset_context(std::shared_ptr<LlamaModel> m) {
    m->n_past = 0; m->n_consumed = 0; m->embd_inp.clear(); m->generatedMessage.clear(); m->lastGeneratedToken = 0;
    if (m->dev) thk_model_reset_kv(m->dev, 0);
}

// th_eval_gpu (th-llama.cpp:464-660) + sync_finish_compute's sampling (:662-727): evaluates
// n_tokens tokens starting at n_past (fed one at a time on the device, as the reference does with
// kAllowedSubsequentBatchSize = 1), reads the last token's logits back and samples.
tk_llama_token th_eval(thk_ctx* ctx, std::shared_ptr<LlamaModel> m, const tk_llama_token* tokens, int n_tokens, int n_past) {
    (void)ctx;
    if (!m || !m->dev) return -1;
    m->logits.resize((size_t)m->n_vocab);
    std::vector<int32_t> ids(tokens, tokens + n_tokens);
    static const std::vector<tk_llama_token> none;
    const SamplerParams& sp = m->sampler;
    const std::vector<tk_llama_token>& last = sp.use_last_n_tokens ? m->last_n_tokens : none;
    // Stochastic sampling without the 4 * n_vocab-byte read-back (the reference maps all logits after every token,
    // th-llama.cpp:686-724): the device selects the few raw logits that can reach the sampler's top_k (thk_model_logits_topk) and
    // the unchanged softmax / top-p / discrete_distribution runs on those.  Ties that would make the result depend on
    // std::partial_sort's internals fall back to the full vector, so the draws are the reference's in every case.
    // optional section timing (LlamaModel::collectStats): tick() returns the seconds since the previous tick
    const bool timed = m->collectStats;
    double t_last = timed ? get_time_seconds() : 0;
    auto tick = [&]() { if (!timed) return 0.0; const double now = get_time_seconds(), d = now - t_last; t_last = now; return d; };
    LlamaModel::LoopStats& stt = m->stats;
    if (m->deviceTopK && sp.temp > 0 && sp.top_k > 0 && sp.top_k < m->n_vocab && m->n_vocab <= 32768) {
        const int K = llama_topk_candidates_needed(m->n_vocab, last, sp.top_k, sp.repeat_penalty);
        if (K <= 1024) {
            std::vector<float> cv((size_t)K); std::vector<int32_t> ci((size_t)K);
            int rc = thk_model_eval_topk(m->dev, 0, ids.data(), n_tokens, n_past, K, cv.data(), ci.data());   // the step and the selection: one stream round trip
            stt.eval_s += tick(); stt.n_eval += 1; stt.n_topk += 1;
            tk_llama_token tok = -1;
            if (rc == THK_OK && llama_sample_from_topk(m->rng, m->n_vocab, last, sp.top_k, sp.top_p, sp.temp, sp.repeat_penalty, cv.data(), ci.data(), K, &tok)) {
                stt.draw_s += tick();
                return tok;
            }
            if (rc == THK_OK) rc = thk_model_read_logits(m->dev, 0, m->logits.data());
            stt.readback_s += tick(); stt.n_readback += 1;
            if (rc != THK_OK) { report_error(*m, std::string("th_eval failed: ") + thk_last_error(m->ctx)); return -1; }
            const tk_llama_token t2 = llama_sample_top_p_top_k(m->rng, m->n_vocab, last, sp.top_k, sp.top_p, sp.temp, sp.repeat_penalty, m->logits);
            stt.draw_s += tick();
            return t2;
        }
    }
    const int rc = thk_model_eval(m->dev, 0, ids.data(), n_tokens, n_past, nullptr, m->logits.data());
    stt.eval_s += tick(); stt.n_eval += 1; stt.n_readback += 1;          // the step and its 4 n_vocab-byte read-back are one call here
    if (rc != THK_OK) {
        report_error(*m, std::string("th_eval failed: ") + thk_last_error(m->ctx));
        return -1;                                  // callers stop on a negative token (the reference returns 0 and carries on)
    }
    const tk_llama_token t3 = llama_sample_top_p_top_k(m->rng, m->n_vocab, last, sp.top_k, sp.top_p, sp.temp, sp.repeat_penalty, m->logits);
    stt.draw_s += tick();
    return t3;
}

// Greedy generation without a per-token host round trip: the reference drains the GPU and maps 128 KB
// of logits after every token (th-llama.cpp:662-727) only to take their arg-max when temp <= 0.  Here the remaining prompt
// tokens are evaluated without any read-back, then the device-resident loop (greedy pick on the GPU, graph replays) runs in
// chunks and only the 4-byte token ids come back; onNewToken fires per token exactly as in the eval path.
static bool greedy_device_loop(std::shared_ptr<LlamaModel> m, int n_ctx, int64_t step) {
    int n_prompt_left = (int)m->embd_inp.size() - m->n_consumed;
    tk_llama_token cur;
    if (n_prompt_left > 0) {
        // as the token loop (its n_past < n_ctx guard): a prompt longer than the context is consumed up to n_ctx, then nothing is generated
        const bool fits = m->n_past + n_prompt_left <= n_ctx;
        if (!fits) n_prompt_left = std::max(0, n_ctx - m->n_past);
        if (n_prompt_left == 0) return true;
        std::vector<int32_t> ids(m->embd_inp.begin() + m->n_consumed, m->embd_inp.begin() + m->n_consumed + n_prompt_left);
        for (int32_t id : ids) { m->last_n_tokens.erase(m->last_n_tokens.begin()); m->last_n_tokens.push_back(id); }
        if (thk_model_seq_set(m->dev, 0, ids[0], m->n_past) != THK_OK) return false;       // clears the device token log
        if (thk_model_eval(m->dev, 0, ids.data(), n_prompt_left, m->n_past, nullptr, nullptr) != THK_OK) return false;
        int32_t last = -1;                                             // greedy continuation of the last prompt token, read directly:
        if (thk_model_seq_last_token(m->dev, 0, &last) != THK_OK || last < 0) return false;   // the device log only holds the first 4096 picks
        cur = last;
        m->n_consumed += n_prompt_left; m->n_past += n_prompt_left; step += n_prompt_left;
        if (!fits) return true;
    } else {
        cur = m->lastGeneratedToken;                                   // the prompt went through thk_model_prefill; its pick is already emitted
        step += 0;
    }
    auto emit = [&](tk_llama_token t) {
        m->lastGeneratedToken = t;
        if (t == tk_llama_token_eos()) return false;
        const char* str = tk_llama_token_to_str(m, t);
        if (str) { m->generatedMessage += str; if (m->onNewToken) m->onNewToken(str, m->generatedMessage); }
        m->last_n_tokens.erase(m->last_n_tokens.begin());
        m->last_n_tokens.push_back(t);
        return true;
    };
    const int64_t limit = std::min<int64_t>(kMaxOutputTokens, m->stepLimit);
    if (m->collectStats && n_prompt_left > 0) m->stats.step_end.insert(m->stats.step_end.end(), (size_t)n_prompt_left, get_time_seconds());
    if (n_prompt_left > 0 && !emit(cur)) return true;
    while (step < limit && m->n_past < n_ctx) {
        const int chunk = (int)std::min<int64_t>(std::min<int64_t>(8, limit - step), n_ctx - m->n_past);
        if (thk_model_seq_set(m->dev, 0, cur, m->n_past) != THK_OK) return false;
        if (thk_model_decode_steps(m->dev, 0, chunk, 1) != THK_OK) return false;
        int32_t toks[8], n_log = 0, pos = 0;
        if (thk_model_seq_get(m->dev, 0, toks, chunk, &n_log, &pos) != THK_OK || n_log != chunk) return false;
        if (m->collectStats) m->stats.step_end.insert(m->stats.step_end.end(), (size_t)chunk, get_time_seconds());
        for (int i = 0; i < chunk; ++i) {
            m->n_past += 1; step += 1;
            if (!emit(toks[i])) return true;                           // EOS: tokens the device produced beyond it are discarded
        }
        cur = toks[chunk - 1];
    }
    return true;
}

// do_inference (th-llama.cpp:111-168) + sync_continue_inference (:199-238): prepend ' ' on a fresh
// context, tokenize with BOS, feed the prompt one token per step, then feed back the sampled
// token.  Unlike the reference the loop stops on EOS (its check reads a vector only the async
// path fills, Q3) and errors surface through onError.
void do_inference(thk_ctx* ctx, std::shared_ptr<LlamaModel> m, std::string prompt) {
    if (!m || !m->dev) return;
    if (m->n_past >= kMaxOutputTokens) {
        report_error(*m, "Maximum context reached (" + std::to_string(kMaxContext) + "). Please reset context by typing '[cmd] reset'");
        return;
    }
    if (m->n_past == 0) prompt.insert(0, 1, ' ');
    m->embd_inp = tk_llama_tokenize(m, prompt, true);
    m->n_consumed = 0;
    if ((int)m->embd_inp.size() > kMaxContext - 4) {
        report_error(*m, "prompt is too long (" + std::to_string(m->embd_inp.size()) + " tokens, max " + std::to_string(kMaxContext - 4) + ")");
        return;
    }
    if (m->last_n_tokens.empty()) m->last_n_tokens.assign(kMaxContext, 0);
    m->generatedMessage.clear();
    const int n_ctx = std::min<int>(m->n_ctx, kMaxContext);
    const int64_t limit = std::min<int64_t>(kMaxOutputTokens, m->stepLimit);
    if (m->collectStats) m->stats.t_begin = get_time_seconds();
    int64_t step = 0;
    const int n_prompt = (int)m->embd_inp.size();
    if (m->prefillPrompt && n_prompt >= 2 && m->n_past + n_prompt <= n_ctx && n_prompt <= limit) {
        // Batched prompt ingestion: one thk_model_prefill call replaces n_prompt steps of the loop below: same KV rows,
        // logits of the last prompt token, every prompt token pushed through last_n_tokens.  The sampler's random stream
        // is advanced by one draw per earlier prompt token (the loop samples after every token and throws the result
        // away while the prompt is consumed).  With temp > 0 this reproduces the token-by-token text EXCEPT when a
        // discarded step would have sampled from a single candidate (top_k == 1, or top_p cutting to one entry):
        // libstdc++'s discrete_distribution draws nothing for one weight, so the streams differ from there on.
        // Greedy sampling (temp <= 0) is exactly equivalent.
        for (int i = 0; i < n_prompt; ++i) { m->last_n_tokens.erase(m->last_n_tokens.begin()); m->last_n_tokens.push_back(m->embd_inp[i]); }
        m->logits.resize((size_t)m->n_vocab);
        std::vector<int32_t> ids(m->embd_inp.begin(), m->embd_inp.end());
        const int rc = thk_model_prefill(m->dev, 0, ids.data(), n_prompt, m->n_past, m->logits.data());
        if (rc != THK_OK) {
            report_error(*m, std::string("prompt prefill failed: ") + thk_last_error(m->ctx));
            return;
        }
        const SamplerParams& sp = m->sampler;
        if (sp.temp > 0) {
            std::discrete_distribution<> two({1.0, 1.0});               // any distribution with > 1 outcome consumes one canonical draw
            for (int i = 0; i < n_prompt - 1; ++i) (void)two(m->rng);
        }
        static const std::vector<tk_llama_token> none;
        m->lastGeneratedToken = llama_sample_top_p_top_k(m->rng, m->n_vocab, sp.use_last_n_tokens ? m->last_n_tokens : none, sp.top_k, sp.top_p,
                                                         sp.temp, sp.repeat_penalty, m->logits);
        m->n_consumed = n_prompt; m->n_past += n_prompt; step = n_prompt;
        if (m->collectStats) m->stats.step_end.insert(m->stats.step_end.end(), (size_t)n_prompt, get_time_seconds());
        if (m->lastGeneratedToken != tk_llama_token_eos()) {
            const char* str = tk_llama_token_to_str(m, m->lastGeneratedToken);
            if (str) {
                m->generatedMessage += str;
                if (m->onNewToken) m->onNewToken(str, m->generatedMessage);
            }
            m->last_n_tokens.erase(m->last_n_tokens.begin());
            m->last_n_tokens.push_back(m->lastGeneratedToken);
        } else {
            step = limit;                                               // EOS straight after the prompt
        }
    }
    if (m->greedyDeviceLoop && m->sampler.temp <= 0 && step < limit && m->n_past < n_ctx) {
        if (!greedy_device_loop(m, n_ctx, step)) report_error(*m, std::string("greedy decode loop failed: ") + thk_last_error(m->ctx));
        if (m->onInferenceComplete) m->onInferenceComplete(m->generatedMessage);
        return;
    }
    for (; step < limit && m->n_past < n_ctx; ++step) {
        tk_llama_token in;
        const bool from_prompt = m->n_consumed < (int)m->embd_inp.size();
        if (from_prompt) {
            in = m->embd_inp[m->n_consumed++];
            m->last_n_tokens.erase(m->last_n_tokens.begin());
            m->last_n_tokens.push_back(in);
        } else {
            in = m->lastGeneratedToken;
        }
        m->lastGeneratedToken = th_eval(ctx, m, &in, 1, m->n_past);
        if (m->collectStats) m->stats.step_end.push_back(get_time_seconds());
        if (m->lastGeneratedToken < 0) break;                           // evaluation failed (already reported): do not feed garbage back
        m->n_past += 1;
        if (m->n_consumed < (int)m->embd_inp.size()) continue;          // still consuming the prompt
        if (m->lastGeneratedToken == tk_llama_token_eos()) break;
        const char* str = tk_llama_token_to_str(m, m->lastGeneratedToken);
        if (str) {
            m->generatedMessage += str;
            if (m->onNewToken) m->onNewToken(str, m->generatedMessage);
        }
        m->last_n_tokens.erase(m->last_n_tokens.begin());
        m->last_n_tokens.push_back(m->lastGeneratedToken);
    }
    if (m->onInferenceComplete) m->onInferenceComplete(m->generatedMessage);
}

// ------------------------------------------------------------------ sampler (th-llama.cpp:802-907)
// softmax over the (already top-k'd, value-descending) candidates, nucleus cut, one discrete_distribution draw (th-llama.cpp:858-907)
static tk_llama_token sample_from_candidates(std::mt19937& rng, std::vector<std::pair<float, tk_llama_token>>& cand, float top_p) {
    float maxl = -std::numeric_limits<float>::infinity();
    for (auto& kv : cand) maxl = std::max(maxl, kv.first);
    std::vector<float> probs;
    probs.reserve(cand.size());
    double sum = 0.0;
    for (auto& kv : cand) { const float p = expf(kv.first - maxl); probs.push_back(p); sum += p; }
    for (auto& p : probs) p /= sum;
    if (top_p < 1.0) {
        double cum = 0.0;
        for (int i = 0; i < (int)probs.size(); ++i) {
            cum += probs[i];
            if (cum >= top_p) { probs.resize(i + 1); cand.resize(i + 1); break; }
        }
        cum = 1.0 / cum;
        for (auto& p : probs) p *= cum;
    }
    std::discrete_distribution<> dist(probs.begin(), probs.end());
    return cand[dist(rng)].second;
}

tk_llama_token llama_sample_top_p_top_k(std::mt19937& rng, int n_vocab, const std::vector<tk_llama_token>& last_n_tokens, int top_k,
                                        float top_p, float temp, float repeat_penalty, const std::vector<float>& logits) {
    const float* pl = logits.data() + logits.size() - n_vocab;
    if (temp <= 0) {   // greedy: first index attaining the maximum
        int best = 0;
        for (int i = 1; i < n_vocab; ++i) if (pl[i] > pl[best]) best = i;
        return best;
    }
    std::vector<std::pair<float, tk_llama_token>> cand;
    cand.reserve(n_vocab);
    const float scale = 1.0f / temp;
    for (int i = 0; i < n_vocab; ++i) {
        float v = pl[i] * scale;
        if (std::find(last_n_tokens.begin(), last_n_tokens.end(), i) != last_n_tokens.end())   // CTRL repetition penalty
            v = pl[i] < 0.0f ? pl[i] * scale * repeat_penalty : pl[i] * scale / repeat_penalty;
        cand.emplace_back(v, i);
    }
    if (top_k > 0 && top_k < n_vocab) {
        std::partial_sort(cand.begin(), cand.begin() + top_k, cand.end(),
                          [](const std::pair<float, tk_llama_token>& a, const std::pair<float, tk_llama_token>& b) { return a.first > b.first; });
        cand.resize(top_k);
    }
    return sample_from_candidates(rng, cand, top_p);
}

// Raw-logit candidates the device must supply so that the top_k of the PENALISED, temperature-scaled values is among them: the
// penalty only ever lowers a value (l * s / p for l >= 0, l * s * p for l < 0, p >= 1), so an entry outside the top (top_k + P + 1)
// raw logits - P = distinct penalised ids - has at least top_k + 1 unpenalised entries strictly above it and cannot be one of the
// top_k + 1 scaled values.  (The + 1 lets the caller see whether the top_k boundary is a tie.)
int llama_topk_candidates_needed(int n_vocab, const std::vector<tk_llama_token>& last_n_tokens, int top_k, float repeat_penalty) {
    if (top_k <= 0 || top_k >= n_vocab || repeat_penalty < 1.0f) return n_vocab;
    std::vector<tk_llama_token> pen;
    for (tk_llama_token t : last_n_tokens) if (t >= 0 && t < n_vocab) pen.push_back(t);
    std::sort(pen.begin(), pen.end());
    const int P = (int)(std::unique(pen.begin(), pen.end()) - pen.begin());
    return std::min(n_vocab, top_k + P + 1);
}

// The same draw as llama_sample_top_p_top_k from the n_cand LARGEST raw logits (value descending, ties by ascending id: what
// thk_model_logits_topk returns) instead of all n_vocab.  Returns false WITHOUT touching the generator when the result could
// depend on how std::partial_sort orders equal values (a tie among the first top_k + 1 scaled values) or when too few candidates
// were supplied: the caller then reads the whole vector back and takes the reference path, so every draw is the reference's.
bool llama_sample_from_topk(std::mt19937& rng, int n_vocab, const std::vector<tk_llama_token>& last_n_tokens, int top_k, float top_p, float temp,
                            float repeat_penalty, const float* cand_logits, const int32_t* cand_ids, int n_cand, tk_llama_token* out) {
    if (temp <= 0 || top_k <= 0 || top_k >= n_vocab) return false;
    if (n_cand < llama_topk_candidates_needed(n_vocab, last_n_tokens, top_k, repeat_penalty)) return false;
    std::vector<std::pair<float, tk_llama_token>> cand;
    cand.reserve(n_cand);
    const float scale = 1.0f / temp;
    for (int j = 0; j < n_cand; ++j) {
        const float l = cand_logits[j];
        const tk_llama_token i = cand_ids[j];
        float v = l * scale;
        if (std::find(last_n_tokens.begin(), last_n_tokens.end(), i) != last_n_tokens.end())
            v = l < 0.0f ? l * scale * repeat_penalty : l * scale / repeat_penalty;
        cand.emplace_back(v, i);
    }
    std::stable_sort(cand.begin(), cand.end(), [](const std::pair<float, tk_llama_token>& a, const std::pair<float, tk_llama_token>& b) { return a.first > b.first; });
    for (int j = 0; j < top_k && j + 1 < (int)cand.size(); ++j)
        if (cand[j].first == cand[j + 1].first) return false;          // partial_sort's order of equal values is its own business: take the reference path
    cand.resize(top_k);
    *out = sample_from_candidates(rng, cand, top_p);
    return true;
}
tk_llama_token llama_sample_top_p_top_k(std::shared_ptr<LlamaModel> m, const std::vector<tk_llama_token>& last_n_tokens, int top_k,
                                        float top_p, float temp, float repeat_penalty, std::vector<float>& logits) {
    return llama_sample_top_p_top_k(m->rng, m->n_vocab, last_n_tokens, top_k, top_p, temp, repeat_penalty, logits);
}

// ------------------------------------------------------------------ tokenizer (th-llama.cpp:910-1108)
// SentencePiece-style BPE: split into UTF-8 characters, repeatedly merge the adjacent pair whose
// concatenation is the best-scoring vocabulary entry (ties: leftmost), then map the surviving
// pieces to ids, falling back to byte tokens (id = byte + 3).
namespace {
struct Piece { int prev, next; size_t off, len; };
struct Merge { int left, right; float score; size_t len; };
struct MergeOrder {
    bool operator()(const Merge& a, const Merge& b) const { return a.score < b.score || (a.score == b.score && a.left > b.left); }
};
size_t utf8_char_len(unsigned char c) {
    static const size_t t[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 4};
    return t[c >> 4];
}
}  // namespace

std::vector<tk_llama_token> tk_llama_tokenize(const LlamaVocab& vocab, const std::string& text, bool add_bos) {
    std::vector<tk_llama_token> out;
    if (text.empty()) return out;
    if (add_bos) out.push_back(tk_llama_token_bos());

    std::vector<Piece> pcs;
    for (size_t off = 0; off < text.size();) {
        const size_t n = std::min(text.size() - off, utf8_char_len((unsigned char)text[off]));
        pcs.push_back({(int)pcs.size() - 1, 0, off, n});
        off += n;
        pcs.back().next = off == text.size() ? -1 : (int)pcs.size();
    }
    std::priority_queue<Merge, std::vector<Merge>, MergeOrder> work;
    auto propose = [&](int l, int r) {
        if (l < 0 || r < 0) return;
        const std::string cat = text.substr(pcs[l].off, pcs[l].len + pcs[r].len);
        auto it = vocab.token_to_id.find(cat);
        if (it == vocab.token_to_id.end() || (size_t)it->second >= vocab.id_to_token.size()) return;
        work.push({l, r, vocab.id_to_token[it->second].score, cat.size()});
    };
    for (int i = 1; i < (int)pcs.size(); ++i) propose(i - 1, i);
    while (!work.empty()) {
        const Merge mg = work.top();
        work.pop();
        Piece& L = pcs[mg.left];
        Piece& R = pcs[mg.right];
        if (L.len == 0 || R.len == 0 || L.len + R.len != mg.len) continue;   // stale proposal
        L.len += R.len; R.len = 0;
        L.next = R.next;
        if (R.next >= 0) pcs[R.next].prev = mg.left;
        propose(L.prev, mg.left);
        propose(mg.left, L.next);
    }
    for (int i = 0; i != -1; i = pcs[i].next) {
        const std::string piece = text.substr(pcs[i].off, pcs[i].len);
        auto it = vocab.token_to_id.find(piece);
        if (it != vocab.token_to_id.end()) out.push_back(it->second);
        else for (unsigned char ch : piece) out.push_back((tk_llama_token)ch + 3);
    }
    return out;
}
std::vector<tk_llama_token> tk_llama_tokenize(std::shared_ptr<LlamaModel> m, const std::string& text, bool add_bos) {
    return tk_llama_tokenize(m->vocab, text, add_bos);
}
const char* tk_llama_token_to_str(std::shared_ptr<LlamaModel> m, tk_llama_token token) {
    if (token < 0 || token >= (tk_llama_token)m->vocab.id_to_token.size()) return nullptr;
    return m->vocab.id_to_token[token].tok.c_str();
}

}  // namespace th
