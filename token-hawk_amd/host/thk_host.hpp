// thk_host.hpp — C++ host layer: TokenHawk's public API shape (namespace th) re-created on
// top of the libthk C-ABI (include/thk.h), so cli-style callers compile with
// `WGPUDevice, WGPUQueue` replaced by one `thk_ctx*`.
//
// Mirrors (same names, argument meaning, error behaviour where sane):
//   th.hpp:20-148        TensorType, TensorShape, TensorBuffer
//   th.hpp:291-294       ggml_compute_fp16_to_fp32 / fp32_to_fp16
//   th-llama.hpp:24-27   build_pipelines_llama (no-op here), tk_llama_token
//   th-llama.hpp:87-179  LlamaVocab, LlamaModel (device state is a thk_model*)
//   th-llama.hpp:225     do_inference
//   th-llama-loader.hpp  load_llama_file, load_header, load_weights, post_load_init_model
// Deliberately NOT mirrored: Q3 (sync loop never sees EOS), Q6
// (assert n_ff==11008), Q8 (leaks), the dead chunked "-d" format, the WebGPU pipeline/bind-group
// types.  Errors are reported through LlamaModel::onError and return values, never by
// assert(false) + silent continue.
#pragma once

#include <stdint.h>
#include <functional>
#include <memory>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/thk.h"

namespace th {

enum TensorType { TensorType_Unknown, TensorType_F16, TensorType_F32 };
std::string get_TensorType_name(TensorType dt);
size_t get_TensorType_size(TensorType dt);

// 4-D shape, 0 == "dimension absent" (th.hpp:37-77).
struct TensorShape {
    int64_t l{}, b{}, r{}, c{};
    int64_t get_total_num_elements() const;
    std::string to_string() const;
    void canonicalize();   // l==1 -> 0, b==1 -> 0, r==0 -> 1
};
inline bool operator==(const TensorShape& x, const TensorShape& y) { return x.l == y.l && x.b == y.b && x.r == y.r && x.c == y.c; }

// Move-only owner of one device allocation (th.hpp:83-148), optional host copy.
struct TensorBuffer {
    TensorBuffer() = default;
    TensorBuffer(TensorShape shape, TensorType type, thk_ctx* ctx = nullptr);
    TensorBuffer(const void* data, TensorShape shape, TensorType type, bool backup, thk_ctx* ctx = nullptr);
    ~TensorBuffer() { free_buffers(); }
    TensorBuffer(const TensorBuffer&) = delete;
    TensorBuffer& operator=(const TensorBuffer&) = delete;
    TensorBuffer(TensorBuffer&& o) noexcept { *this = std::move(o); }
    TensorBuffer& operator=(TensorBuffer&& o) noexcept;

    size_t get_size_bytes() const;
    bool is_valid() const { return shape.get_total_num_elements() > 0 && (gpu != nullptr || !cpuBackup.empty()); }
    bool upload_data_to_gpu(const void* data);
    bool download(void* out) const;
    void reset_shape() { shape = originalShape; }
    void* device_ptr() const { return gpu ? thk_buf_ptr(gpu) : nullptr; }
    void free_buffers();

    TensorShape shape{}, originalShape{};
    TensorType type = TensorType_Unknown;
    thk_ctx* ctx = nullptr;
    thk_buf* gpu = nullptr;
    std::vector<uint8_t> cpuBackup;
};

typedef uint16_t ggml_fp16_t;
ggml_fp16_t ggml_compute_fp32_to_fp16(float f);
float ggml_compute_fp16_to_fp32(ggml_fp16_t h);

typedef int tk_llama_token;

struct LlamaVocab {
    using id = int32_t;
    using token = std::string;
    struct token_score { token tok; float score; };
    std::unordered_map<token, id> token_to_id;
    std::vector<token_score> id_to_token;
};

// Sampling parameters th_eval_gpu hard-codes (th-llama.cpp:719-722).  The reference passes an
// empty last_n_tokens to the sampler (Q2), so the repetition penalty never applies; that is the
// default here too.
struct SamplerParams {
    int32_t top_k = 40;
    float top_p = 0.95f;
    float temp = 0.80f;
    float repeat_penalty = 1.10f;
    bool use_last_n_tokens = false;
};

struct LlamaModel {
    std::mt19937 rng{};
    // hyper-parameters; defaults are LLaMA-7B (th-llama.hpp:103-112)
    int32_t n_vocab = 32000, n_ctx = 512, n_embd = 4096, n_mult = 256, n_head = 32, n_layer = 32, n_rot = 64, n_batch = 8, f16 = 1;

    thk_ctx* ctx = nullptr;       // borrowed
    thk_model* dev = nullptr;     // owned: weights, KV caches, working buffers, decode graphs
    LlamaVocab vocab{};
    SamplerParams sampler{};
    int lmhead_mode = THK_LMHEAD_CORRECT;   // THK_LMHEAD_FAITHFUL reproduces defect Q1
    // false (default): the prompt is fed one token per step, as the reference does (kAllowedSubsequentBatchSize = 1,
    // th-llama.cpp:15).  true: the whole prompt goes through thk_model_prefill (MFMA GEMMs) in one call; the sampler's
    // random stream is advanced by the draws the token loop would have made, so the generated text is the same.
    bool prefillPrompt = false;
    // true (default): with greedy sampling (temp <= 0) generation runs in the device-resident loop (thk_model_decode_steps,
    // 4-byte token read-backs in chunks of 8) instead of one blocking 128 KB logits read-back per token; same text.
    bool greedyDeviceLoop = true;
    bool deviceTopK = true;           // temp > 0: top-k candidates selected on the device instead of reading all n_vocab logits back per token
    // Steps (prompt tokens + generated tokens) one do_inference call may take: the reference's kMaxOutputTokens (th-llama.cpp:17)
    // unless the embedder asks for less (bench.py's host_api figure generates a fixed number of tokens).
    int64_t stepLimit = 500;
    // Optional per-call timing of the token loop (off by default: two clock reads per section).  Seconds, accumulated over the
    // calls since the last clear(): th_eval's sections and the wall-clock time at which every step of do_inference ended.
    struct LoopStats {
        double eval_s = 0, topk_s = 0, readback_s = 0, draw_s = 0;      // device step (replay + sync) | top-k kernel + k x 8 B read-back | 4 n_vocab B read-back | host sampler
        int64_t n_eval = 0, n_topk = 0, n_readback = 0;
        double t_begin = 0;                                             // do_inference entry
        std::vector<double> step_end;                                   // one entry per step (the greedy device loop stamps a chunk's steps together)
        void clear() { *this = LoopStats{}; }
    };
    bool collectStats = false;
    LoopStats stats;

    std::function<void(std::string /*token*/, std::string /*messageSoFar*/)> onNewToken;
    std::function<void(std::string /*fullMessage*/)> onInferenceComplete;
    std::function<void(std::string /*terminate_reason*/)> onError;

    // loading state
    bool loadFailed = false;
    int64_t numTensorsLoaded = 0;
    std::vector<std::string> loadedNames;

    // active processing state (th-llama.hpp:168-178)
    std::vector<tk_llama_token> embd_inp{};
    int n_past = 0;
    int n_consumed = 0;
    tk_llama_token lastGeneratedToken{};
    std::string generatedMessage;
    std::vector<tk_llama_token> last_n_tokens{};
    std::vector<float> logits;    // last evaluated token's logits (host)

    ~LlamaModel();
};

static const int64_t kMaxOutputTokens = 500;   // th-llama.cpp:17
static const int kMaxContext = 512;            // th-llama.cpp:19

// ---- loader (th-llama-loader.hpp:8-16).  ggjt v1 only, f16/f32 tensors only.
std::shared_ptr<LlamaModel> load_llama_file(thk_ctx* ctx, const std::string& filename, int lmhead_mode = THK_LMHEAD_CORRECT);
bool load_header(LlamaModel* m, const void* data, int64_t dataSize, int64_t* consumed = nullptr);
bool load_weights(LlamaModel* m, thk_ctx* ctx, const void* data, int64_t dataSize, int64_t numElementsInFile, int64_t originalFileOffset);
bool post_load_init_model(thk_ctx* ctx, std::shared_ptr<LlamaModel> m);
void build_pipelines_llama(thk_ctx* ctx, std::shared_ptr<LlamaModel> m);   // kernels are AOT-compiled: nothing to build

// Parsed view of one tensor record (used by load_weights and by the parse-only test hooks).
struct GgjtTensorInfo {
    std::string name;
    TensorType type = TensorType_Unknown;
    TensorShape shape{};
    int64_t ne0 = 0, ne1 = 1;       // ggml order: ne0 = columns
    int64_t data_offset = 0;        // offset of the payload inside the record buffer
    int64_t data_bytes = 0;
    int64_t record_bytes = 0;       // header + padding + payload
};
bool parse_tensor_record(const void* data, int64_t dataSize, int64_t originalFileOffset, GgjtTensorInfo* out, std::string* err);

// ---- inference (th-llama.hpp:225, th-llama.cpp:28-33, :111-238)
void do_inference(thk_ctx* ctx, std::shared_ptr<LlamaModel> m, std::string prompt);
tk_llama_token th_eval(thk_ctx* ctx, std::shared_ptr<LlamaModel> m, const tk_llama_token* tokens, int n_tokens, int n_past);
void reset_context(std::shared_ptr<LlamaModel> m);    // web "[cmd] reset" (web/main.cpp:164-170)

// ---- tokenizer / sampler (th-llama.cpp:35-47, :802-1108)
std::vector<tk_llama_token> tk_llama_tokenize(const LlamaVocab& vocab, const std::string& text, bool add_bos);
std::vector<tk_llama_token> tk_llama_tokenize(std::shared_ptr<LlamaModel> m, const std::string& text, bool add_bos);
const char* tk_llama_token_to_str(std::shared_ptr<LlamaModel> m, tk_llama_token token);
inline tk_llama_token tk_llama_token_bos() { return 1; }
inline tk_llama_token tk_llama_token_eos() { return 2; }
tk_llama_token llama_sample_top_p_top_k(std::mt19937& rng, int n_vocab, const std::vector<tk_llama_token>& last_n_tokens, int top_k,
                                        float top_p, float temp, float repeat_penalty, const std::vector<float>& logits);
// The same draw from the few largest raw logits (thk_model_logits_topk) instead of all n_vocab; false = take the full path (see thk_llama.cpp)
int llama_topk_candidates_needed(int n_vocab, const std::vector<tk_llama_token>& last_n_tokens, int top_k, float repeat_penalty);
bool llama_sample_from_topk(std::mt19937& rng, int n_vocab, const std::vector<tk_llama_token>& last_n_tokens, int top_k, float top_p, float temp,
                            float repeat_penalty, const float* cand_logits, const int32_t* cand_ids, int n_cand, tk_llama_token* out);
tk_llama_token llama_sample_top_p_top_k(std::shared_ptr<LlamaModel> m, const std::vector<tk_llama_token>& last_n_tokens, int top_k,
                                        float top_p, float temp, float repeat_penalty, std::vector<float>& logits);

// ---- timing stats (th.cpp:23-87)
double get_time_seconds();
std::string descriptive_stats(std::vector<double> data, const std::string& unit);

}  // namespace th
