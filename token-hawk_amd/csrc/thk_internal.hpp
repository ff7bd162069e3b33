// thk_internal.hpp — shared by the host-side translation units that implement include/thk.h:
//   thk_ctx.cpp            context, tunables, buffers (TensorBuffer's GPU half, th.cpp:150-229)
//   thk_ops.cpp            one operator per reference kernel (the 16 cmdbuf_* encoders, th.cpp:617-4351)
//   thk_model.cpp          model level: objects, tensors, finalize, thk_model_eval, sequence accessors
//   thk_model_step.cpp     the decode step (th_eval_gpu's body, th-llama.cpp:464-660) as hipGraph replays, step-level API
//   thk_model_prefill.cpp  MFMA prompt prefill; thk_model_engine.cpp  the optional one-launch engine's program
//   thk_pp.cpp / thk_peer.hip  stage-to-stage transports
// Internal; not part of the ABI.
#pragma once
#include "../../include/thk.h"
#include "thk_kernels.hpp"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <string>
#include <array>
#include <vector>

using namespace thk;

// ---------------------------------------------------------------- objects
struct thk_buf {
    void* ptr = nullptr;
    size_t size = 0;
};

struct thk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err = "";
    std::map<std::string, int64_t> tun;
    int n_cu = 256;
    size_t hbm_bytes = 0;
    std::string dev_name;
    void* scratch = nullptr;        // operator-API scratch (attention partials, arg-max keys)
    size_t scratch_bytes = 0;
    float* rope_tab = nullptr;      // operator-API RoPE table
    size_t rope_tab_floats = 0;
    unsigned long long* pinned_keys = nullptr;       // host-mapped page the top-k kernel writes its k keys into (thk_model_eval_topk): no copy operation,
    unsigned long long* pinned_keys_dev = nullptr;   // word [1024] is the kernel's "done" stamp: the host polls it instead of synchronising the stream
    unsigned long long topk_epoch = 0;
};

struct LayerW {
    uint16_t *wq = nullptr, *wk = nullptr, *wv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr, *w3 = nullptr;
    float *attention_norm = nullptr, *ffn_norm = nullptr;
};

struct SeqBuf {
    float* kv = nullptr;             // [n_local_layers][2][n_ctx*E] f32 (or binary16 when the model was finalized with kv_f16)
    SeqState* st = nullptr;          // device
    int32_t* gen_log = nullptr;      // device, kGenLogCap
    unsigned long long* clock_log = nullptr;   // device, kGenLogCap: s_memrealtime (100 MHz) at the end of the step that logged gen_log[i]
    float* hidden_in = nullptr;      // device f32[E]
    float* hidden_out = nullptr;     // device f32[E]
    float* logits = nullptr;         // device f32[V] (head stage)
    int32_t* advance = nullptr;      // device flag read by the finishing kernel
    int advance_host = -1;           // last value written
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::map<int, std::pair<hipGraph_t, hipGraphExec_t>> multi;     // n decode steps (2 <= n <= kMaxGraphSteps) in ONE graph (thk_model_prepare_steps / first use)
    std::map<int, unsigned long long> multi_used; unsigned long long multi_clock = 0;   // last use per n: at most kMaxMultiGraphs are kept (LRU)
    int pos_host = 0;                // position the NEXT step evaluates (every change goes through this API, so the host knows it exactly)
    EngOp* eng_ops = nullptr;        // device: this sequence's engine program (cache / hidden-state pointers differ per sequence)
    int eng_n_ops = 0;
};

static const int kGenLogCap = 4096;
static const int kMaxMultiGraphs = 6;     // multi-step graphs kept per sequence (n * ~161 kernel nodes each); the least recently used one is evicted
static const int kMaxGraphSteps = 32;     // longest multi-step graph; a request of n steps replays floor(n / 32) of these and ONE graph of the remainder

struct thk_model {
    thk_ctx* ctx = nullptr;
    thk_hparams hp{};
    int n_ff = 0, l0 = 0, l1 = 0, n_seq = 1;
    uint32_t flags = 0;
    int lm_mode = THK_LMHEAD_CORRECT;
    bool finalized = false;
    std::vector<LayerW> layers;       // local layers
    uint16_t* tok_embeddings = nullptr;
    float* norm = nullptr;
    uint16_t* output = nullptr;
    std::vector<void*> arena_chunks; size_t arena_off = 0, arena_cap = 0;   // working buffers (thk_model.cpp, arena_alloc)
    void* weights_slab = nullptr; size_t weights_slab_bytes = 0;   // one allocation behind every weight pointer above
    std::vector<SeqBuf> seqs;
    // working buffers shared by all sequences (steps run back to back on one stream)
    float *x = nullptr, *q = nullptr, *u = nullptr, *attn_out = nullptr, *part_o = nullptr, *part_ml = nullptr;
    unsigned long long* block_best = nullptr;       // arg-max key per lm-head workgroup of the decode step
    unsigned long long* block_best_aux = nullptr;   // the same for head launches outside the step (prefill)
    size_t block_best_slots = 0;                    // key slots in each of the two arrays
    float* rope_tab = nullptr;        // [n_ctx][D/2][2]
    // launch geometry resolved at finalize
    int nsplit = 4, tc = 128, nt = 1, use_graph = 1;
    int var_qkv = 0, var_wo = 0, var_w13 = 0, var_w2 = 0, var_head = 0;
    int grid_qkv = 0, grid_wo = 0, grid_w13 = 0, grid_w2 = 0, grid_head = 0;
    void* prefill_ws = nullptr; size_t prefill_ws_bytes = 0;
    // tile images of the layer matrices for the prefill GEMM (tunable prefill_packed; built on the first prefill call, one slab for
    // the whole stage = a second copy of the layer weights): pk_w[layer][wq wk wv wo w1 w2 w3]
    void* prefill_pk = nullptr; size_t prefill_pk_bytes = 0;
    int pk_tiles[4] = {0, 0, 0, 0};      // tile rows (qkv, wo, w13, w2) the images were made with; 0 = none
    bool pk_failed = false;              // the slab did not fit once: stay on row-major weights
    std::vector<std::array<const uint16_t*, 7>> pk_w;
    int attn_waves = 8;
    int gain_alias = 0;    // measurement aid (tunable measure_gain_alias)
    int skip_kernel = 0;   // measurement aid (tunable measure_skip_kernel): 1 qkv, 2 attention, 3 wo, 4 w13, 5 w2, 6 lm-head are NOT launched
    // persistent loader/consumer engine (thk_engine.hip): one launch per decode step instead of 5 per layer
    int fold_embed = 1;                  // tunable fold_embed: layer 0's qkv prologue fetches the embedding row (no embed launch)
    int kv_f16 = 0;                      // tunable kv_f16 at finalize: K/V caches stored as binary16 (default 0 = f32, as the reference)
    int engine = 0;                      // resolved at finalize (tunable "engine" and shape eligibility)
    int eng_NS = 0, eng_v0 = 0, eng_v1 = 0, eng_nsplit = 1, eng_tc = 0;
    unsigned long long* eng_gran = nullptr;   // all granule arrays: XG[2][E] | QG[3E] | OG[E] | UG[F] | PG[H*S*(D+2)]
    unsigned* eng_words = nullptr;       // [0] epoch, [32] error word
    unsigned long long* eng_trace = nullptr;   // development timeline (tunable engine_trace), [n_cu][n_ops][8]
    unsigned long long* trace_buf = nullptr;   // development timeline of the launch path (thk_model_step_trace, THK_TRACE builds)
    bool trace_on = false;
    int fold_finish = 1;                 // tunable fold_finish: the lm-head launch's last workgroup picks the greedy token (no finish_token launch)
    int attn_tc_dyn = 1;                 // tunable attn_tc_dyn: the context splits partition the LIVE context (computed on the device), not the cache capacity
};

// ---------------------------------------------------------------- model internals shared by thk_model*.cpp
// K / V cache of local layer i (rows of E elements, f32 or binary16): byte arithmetic, typed as float* for the kernel args
static inline float* kcache_of(const thk_model* m, const SeqBuf& sb, int i) {
    const size_t row = (size_t)m->hp.n_ctx * m->hp.n_embd * (m->kv_f16 ? 2 : 4);
    return reinterpret_cast<float*>(reinterpret_cast<char*>(sb.kv) + (size_t)i * 2 * row);
}
static inline float* vcache_of(const thk_model* m, const SeqBuf& sb, int i) {
    const size_t row = (size_t)m->hp.n_ctx * m->hp.n_embd * (m->kv_f16 ? 2 : 4);
    return reinterpret_cast<float*>(reinterpret_cast<char*>(sb.kv) + ((size_t)i * 2 + 1) * row);
}

int set_seq_state(thk_model* m, int seq, int token, int pos, bool reset_gen);   // thk_model.cpp
int step_enqueue(thk_model* m, int seq);                                        // thk_model_step.cpp: one decode step on the ctx stream (eager or under capture)
int step_run(thk_model* m, int seq);                                            //   ... as a replay of the sequence's one-step graph when graphs are on
int step_set_advance(thk_model* m, int seq, int advance);                       //   the device-resident "advance the position" flag
bool engine_plan(thk_model* m);                                                 // thk_model_engine.cpp
int engine_build_program(thk_model* m, SeqBuf& sb);
int check_engine_error(thk_model* m);
int report_pick_timeout(thk_model* m, int seq);                                 // thk_model.cpp: SeqState::pad seen - repair the key slots, clear the word, fail
// ---------------------------------------------------------------- helpers (thk_ctx.cpp)
int fail(thk_ctx* ctx, int code, const char* fmt, ...);
#define HIPCHK(ctx, call)                                                                                  \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) return fail((ctx), THK_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define REQUIRE(ctx, cond, ...) do { if (!(cond)) return fail((ctx), THK_ERR_INVALID, __VA_ARGS__); } while (0)
int64_t tun(thk_ctx* ctx, const char* name);
void default_tunables(thk_ctx* ctx);
struct Geo { int bpc, var; };
Geo auto_geometry(const char* kernel, int n_embd);
int resolve_variant(thk_ctx* ctx, const char* kernel, int n_embd);
int grid_for(thk_ctx* ctx, const char* specific, int n_groups, int n_embd = 4096);
int ensure_scratch(thk_ctx* ctx, size_t bytes);
void build_rope_table(std::vector<float>& tab, int D, int p0, int n);
uint64_t synth_key(const char* name, uint64_t seed);
float synth_scale(float sigma);
// operators shared with the model level (thk_ops.cpp)
int valid_head_dim(int64_t D);
int valid_splits(int64_t s);
void q1_constants(int V, int* split, int* cov);
int topk_to_host(thk_ctx* ctx, const float* logits_dev, int64_t V, int32_t k, float* values_out, int32_t* ids_out);
int topk_enqueue_pinned(thk_ctx* ctx, const float* logits_dev, int64_t V, int32_t k);                 // kernel -> ctx->pinned_keys, no synchronisation
int topk_wait_pinned(thk_ctx* ctx);                                                                   // until the kernel's stamp arrives (spin, then the stream)
void topk_decode_keys(const unsigned long long* keys, int32_t k, float* values_out, int32_t* ids_out);
hipError_t attn_prefill_dispatch(thk_ctx* ctx, const float* q, const float* kc, const float* vc, int n_past, int M, int H, int D, float* out, bool kv_f16 = false);
