/*
 * thk.h — C-ABI of libthk: MI355X (gfx950) native single-token LLaMA decode
 * behind TokenHawk's host API.
 *
 * This is the drop-in boundary.  It replaces everything the
 * reference does below th_eval_gpu / cmdbuf_*: the WebGPU runtime calls
 * (wgpuDeviceCreateBuffer, wgpuQueueWriteBuffer, wgpuCommandEncoder*,
 * wgpuQueueSubmit, wgpuBufferMapAsync …) and the 16 WGSL kernels in th.cpp.
 * Each entry point cites the reference interface it stands in for
 * (/root/reference file:line).  INTEGRATION.md shows the C++ binding a
 * TokenHawk maintainer would add on top of it.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types.
 *   - Every call returns 0 on success or a negative thk_status; the message is
 *     available from thk_last_error(ctx).  No exceptions cross the boundary.
 *     (The reference printf's + assert(false)'s and returns an empty
 *     CommandBuffer, th.cpp:541-608; release builds then continue silently —
 *     this ABI fails loudly instead.)
 *   - A thk_ctx is bound to ONE HIP device and ONE stream.  All ops are
 *     enqueued on that stream in call order and are asynchronous unless stated
 *     otherwise; thk_sync() drains it.  Thread-compatible, not thread-safe.
 *   - "dev" pointers are device addresses valid on the ctx's device (from
 *     thk_buf_ptr, or any hipMalloc'd/torch-owned memory).
 *   - Weights are GGML f16 row-major [R rows (out features), C cols (in
 *     features)]; activations and KV caches are f32, as in the reference.
 */
#ifndef THK_H
#define THK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden */

#define THK_ABI_VERSION 1

typedef struct thk_ctx thk_ctx;
typedef struct thk_buf thk_buf;
typedef struct thk_model thk_model;

typedef enum thk_status {
    THK_OK = 0,
    THK_ERR_INVALID = -1,   /* bad argument / shape the kernels do not support */
    THK_ERR_HIP = -2,       /* a HIP runtime call failed (message has hipGetErrorString) */
    THK_ERR_OOM = -3,
    THK_ERR_STATE = -4,     /* call order violated (e.g. eval before finalize) */
    THK_ERR_NOTFOUND = -5,  /* unknown tensor name */
    THK_ERR_RCCL = -6
} thk_status;

/* TensorType (th.hpp:20-34); values match the ggjt ftype field (loader :18-19). */
typedef enum thk_dtype { THK_F32 = 0, THK_F16 = 1 } thk_dtype;

/* lm-head combine mode: CORRECT sums both K halves for every
 * logit; FAITHFUL reproduces cmdbuf_vector_reduce(...,8) (th-llama.cpp:262,
 * th.cpp:3930-3943, :3992-3996), which skips 160 of every 4000 logits. */
typedef enum thk_lmhead_mode { THK_LMHEAD_CORRECT = 0, THK_LMHEAD_FAITHFUL = 1 } thk_lmhead_mode;

/* ---------------------------------------------------------------- context
 * Replaces WGPUDevice + WGPUQueue (cli/main.cpp:72-105). */
int thk_abi_version(void);
int thk_ctx_create(int device_ordinal, thk_ctx** out);
/* Bind to a caller-owned hipStream_t (e.g. torch's current stream). */
int thk_ctx_create_on_stream(int device_ordinal, void* hip_stream, thk_ctx** out);
int thk_ctx_destroy(thk_ctx* ctx);
int thk_sync(thk_ctx* ctx);                       /* wgpuDeviceTick spin, th-llama.cpp:700-706 */
const char* thk_last_error(thk_ctx* ctx);         /* never NULL */
void* thk_ctx_stream(thk_ctx* ctx);               /* the hipStream_t in use */
/* Device properties the host layer reports (name, CU count, HBM bytes). */
int thk_ctx_device_info(thk_ctx* ctx, char* name, size_t name_cap, int* n_cu, size_t* hbm_bytes);

/* ---------------------------------------------------------------- buffers
 * Replace TensorBuffer's GPU half (th.hpp:83-148, th.cpp:150-229):
 * wgpuDeviceCreateBuffer / wgpuQueueWriteBuffer / CopyBufferToBuffer / MapAsync. */
int thk_buf_alloc(thk_ctx* ctx, size_t bytes, thk_buf** out);   /* zero-initialised */
int thk_buf_free(thk_ctx* ctx, thk_buf* buf);
void* thk_buf_ptr(thk_buf* buf);
size_t thk_buf_size(thk_buf* buf);
int thk_buf_upload(thk_ctx* ctx, thk_buf* dst, size_t dst_off, const void* host, size_t bytes);   /* blocking */
int thk_buf_download(thk_ctx* ctx, thk_buf* src, size_t src_off, void* host, size_t bytes);       /* blocking */
int thk_buf_copy(thk_ctx* ctx, thk_buf* dst, size_t dst_off, thk_buf* src, size_t src_off, size_t bytes);

/* ---------------------------------------------------------------- operators
 * One per starred kernel of SURVEY.md §2b, explicit dims instead of constants
 * baked into generated WGSL.  All pointers are dev pointers. */

/* cmdbuf_vector_mat_mul_trans (th.hpp:404-412, th.cpp:2839-3139):
 * y[r] = sum_c x[c] * f16(W[r,c]), f32 accumulate.  Requires C % 256 == 0. */
int thk_matvec_f16(thk_ctx* ctx, const void* W, int64_t R, int64_t C, const float* x, float* y);

/* cmdbuf_rms_norm (th.hpp:328-333, th.cpp:1153-1296): in place per row,
 * x /= sqrt(mean(x^2) + 1e-6).  Requires N % 256 == 0. */
int thk_rms_norm(thk_ctx* ctx, float* x, int64_t rows, int64_t N);

/* cmdbuf_row_element_multiply (th.hpp:335-341, th.cpp:1298-1449): x[r,c] *= gain[c]. */
int thk_row_element_multiply(thk_ctx* ctx, float* x, const float* gain, int64_t rows, int64_t N);

/* cmdbuf_RoPE (th.hpp:343-349, th.cpp:1452-1616): x viewed [n_tok,H,D], in place,
 * position of token t is n_past + t.  Uses a host-built cos/sin table (libm f32). */
int thk_rope(thk_ctx* ctx, float* x, int64_t n_tok, int64_t H, int64_t D, int64_t n_past);

/* K/V append, wgpuCommandEncoderCopyBufferToBuffer at th-llama.cpp:332-339:
 * kcache[pos,:,:] = k; vcache[pos,:,:] = v; caches f32 [n_ctx,H,D]. */
int thk_kv_append(thk_ctx* ctx, float* kcache, float* vcache, const float* k, const float* v,
                  int64_t pos, int64_t H, int64_t D);

/* Decode attention for one query token over T cached positions.  Replaces the
 * chain transpose x3 + mat_mul(QK^T, scale 1/sqrt(D)) + row_softmax + mat_mul(PV)
 * + transpose (th-llama.cpp:341-397; kernels K8,K9,K10 th.cpp:863-1151, :396-861,
 * :1865-2119).  Reads the caches in place ([n_ctx,H,D]); out is [H*D]. */
int thk_attn_decode(thk_ctx* ctx, const float* q, const float* kcache, const float* vcache,
                    int64_t T, int64_t H, int64_t D, float* out);

/* Causal attention for M query tokens at positions [n_past, n_past + M) over the cached positions
 * (the rows of those M tokens must already be in the caches).  Stands in for the batch branch of
 * build_layer_cmdbuf: mat_mul(QK^T) + cmdbuf_masked_softmax (K14, th.cpp:1619-1863) + mat_mul(PV)
 * (th-llama.cpp:365-404) with the mask the reference intended: query i sees positions
 * <= n_past + i (the reference's own mask ignores n_past).  q and out are [M, H*D]. */
int thk_attn_prefill(thk_ctx* ctx, const float* q, const float* kcache, const float* vcache,
                     int64_t n_past, int64_t M, int64_t H, int64_t D, float* out);

/* cmdbuf_row_softmax (th.hpp:359-365, th.cpp:1865-2119): in place, rows x N. */
int thk_row_softmax(thk_ctx* ctx, float* x, int64_t rows, int64_t N);

/* cmdbuf_addition (th.hpp:367-374), cmdbuf_silu (:396-401),
 * cmdbuf_element_mult_in_place (:386-392). */
int thk_add(thk_ctx* ctx, const float* a, const float* b, float* c, int64_t n);
int thk_silu(thk_ctx* ctx, float* x, int64_t n);
int thk_mul_inplace(thk_ctx* ctx, float* a, const float* b, int64_t n);

/* lm-head: cmdbuf_vector_multi_mat_mul_split_trans + cmdbuf_vector_reduce
 * (th.hpp:427-449, th.cpp:3516-4127) over the UNSPLIT [V,E] f16 matrix (the
 * 256 MB WebGPU buffer limit that forced the split does not exist here). */
int thk_lmhead_f16(thk_ctx* ctx, const void* W, int64_t V, int64_t E, const float* x, float* logits, int mode);

/* Greedy pick, llama_sample_top_p_top_k temp<=0 branch (th-llama.cpp:826-838):
 * smallest index attaining the max.  id_out is a dev int32. */
int thk_argmax(thk_ctx* ctx, const float* logits, int64_t V, int32_t* id_out);
/* The k largest of V device-resident f32 logits, value descending and ties by ascending index (a total order, so the result is
 * unique), selected and sorted on the GPU; only k (value, index) pairs are copied to the host.  V <= 32768, k <= 1024.
 * Replaces the 4 * n_vocab-byte map-read + CPU partial_sort of the stochastic sampler (th-llama.cpp:686-724, :814-857). */
int thk_topk_f32(thk_ctx* ctx, const float* logits, int64_t V, int32_t k, float* values_out, int32_t* ids_out);

/* Embedding row fetch (th-llama.cpp:577-584 + loader :185-195): x = f32(table[token,:]),
 * table f16 [V,E] on device (the reference keeps an f32 copy on the host). */
int thk_embed_f16(thk_ctx* ctx, const void* table, int64_t E, int32_t token, float* x);

/* Prefill GEMM on MFMA (config C3; stands in for cmdbuf_mat_mul with f16 B and
 * transposeB=1 on the batch path, th-llama.cpp:307-311): Y[M,R] = X[M,C] * W[R,C]^T,
 * X,Y f32 row-major, W f16.  X is split into f16 hi+lo parts so the product keeps
 * ~f32 activation precision.  Requires C % 32 == 0. */
int thk_gemm_f16_prefill(thk_ctx* ctx, const void* W, int64_t R, int64_t C, const float* X, int64_t M, float* Y);

/* Synthetic tensor generator (bit-identical to oracle/thk_oracle.c orc_synth_*):
 * fills n elements of a dev buffer from (name, seed, sigma). */
int thk_synth_f16(thk_ctx* ctx, const char* name, uint64_t seed, float sigma, int64_t n, void* out);
int thk_synth_gain_f32(thk_ctx* ctx, const char* name, uint64_t seed, float sigma, int64_t n, float* out);

/* ---------------------------------------------------------------- model
 * Replaces LlamaModel's device state + th_eval_gpu (th-llama.hpp:100-179,
 * th-llama.cpp:464-660) for a contiguous layer range (pipeline stage). */
typedef struct thk_hparams {
    int32_t n_vocab, n_embd, n_mult, n_head, n_layer, n_ctx;   /* th-llama.hpp:103-112 */
} thk_hparams;

/* Stage flags */
#define THK_STAGE_EMBED 1u  /* this stage owns tok_embeddings and starts from a token id */
#define THK_STAGE_HEAD  2u  /* this stage owns norm + output and produces logits / argmax */

/* layers [layer_begin, layer_end) live on this ctx; n_seq independent sequences
 * (each with its own KV caches and position) can be in flight. */
int thk_model_create(thk_ctx* ctx, const thk_hparams* hp, int32_t layer_begin, int32_t layer_end,
                     uint32_t stage_flags, int32_t n_seq, thk_model** out);
int thk_model_destroy(thk_model* m);
int32_t thk_model_n_ff(const thk_model* m);
int32_t thk_model_n_embd(const thk_model* m);
int32_t thk_model_n_ctx(const thk_model* m);

/* Called by the GGML loader in file order (replaces load_weights' TensorBuffer
 * upload, th-llama-loader.cpp:121-265).  name is the ggjt tensor name; ne0 = columns
 * (input features), ne1 = rows (1 for 1-D tensors).  Tensors of layers outside the
 * stage (or embeddings/head on a stage without them) are accepted and ignored. */
int thk_model_set_tensor(thk_model* m, const char* name, int dtype, int64_t ne0, int64_t ne1, const void* host);
/* Same, payload already on the device (e.g. thk_buf_ptr of a buffer the caller uploaded -- the host layer's TensorBuffer,
 * th.hpp:83-148): one stream-ordered device-to-device copy; the source may be freed after thk_sync / thk_buf_free. */
int thk_model_set_tensor_dev(thk_model* m, const char* name, int dtype, int64_t ne0, int64_t ne1, const void* dev_ptr);
/* Reads bytes [offset_bytes, offset_bytes + n_bytes) of a tensor this stage owns back to the host (row-major, as it was set): the inverse of
 * thk_model_set_tensor, for a loader's self-check (the reference keeps a cpuBackup for the same purpose, th.hpp:142).  Blocking. */
int thk_model_get_tensor(thk_model* m, const char* name, int64_t offset_bytes, int64_t n_bytes, void* host_out);
/* Device-side fill of every tensor this stage owns with the synthetic generator. */
int thk_model_fill_synthetic(thk_model* m, uint64_t seed, float sigma);
/* Allocates caches/working buffers, builds the RoPE table and captures the
 * per-sequence decode hipGraphs (replaces post_load_init_model + build_pipelines_llama,
 * th-llama-loader.cpp:330-435, th-llama.cpp:66-76). */
int thk_model_finalize(thk_model* m);
int thk_model_reset_kv(thk_model* m, int32_t seq);          /* web "[cmd] reset", web/main.cpp:164-170 */
int thk_model_set_lmhead_mode(thk_model* m, int mode);      /* default THK_LMHEAD_CORRECT */

/* th_eval_gpu for n_tokens tokens fed one at a time (kAllowedSubsequentBatchSize=1,
 * th-llama.cpp:15,:202): runs this stage for each token at positions n_past..;
 * blocking.  tokens may be NULL on a non-embed stage; hidden_inout (host f32[E],
 * may be NULL) supplies the stage input when there is no embedding and receives the
 * stage output; logits_out (host f32[V], may be NULL) receives the last token's logits
 * on a head stage. */
int thk_model_eval(thk_model* m, int32_t seq, const int32_t* tokens, int32_t n_tokens, int32_t n_past,
                   float* hidden_inout, float* logits_out);

/* Batched prompt prefill (config C3) through the MFMA GEMM path: same contract as
 * thk_model_eval with n_past == 0..; full-model stages (pipeline stages: thk_model_prefill_stage
 * below).  The first call builds the weight tile images (tunable prefill_packed) and the
 * workspace; later calls reuse them. */
int thk_model_prefill(thk_model* m, int32_t seq, const int32_t* tokens, int32_t n_tokens, int32_t n_past,
                      float* logits_out);
/* The same pass for ONE pipeline stage (config C3 x C4): this stage's layers [layer_begin, layer_end) over the prompt rows, so that an
 * N-GPU pipeline ingests a prompt with one MFMA pass per stage instead of n_tokens ring revolutions.  The reference's batch branch is
 * per layer (th-llama.cpp:305-311, :365-404) - nothing in it needs the whole model.  hidden_dev is a caller-owned DEVICE buffer of
 * f32 [n_tokens, n_embd] (thk_buf_alloc, or the target of thk_pp_recv / thk_peer_recv_bulk), used in place:
 *   embedding stage: tokens (host int32[n_tokens]) are the input, hidden_dev is only written;
 *   other stages   : rows [0, n_tokens) of hidden_dev are the input (tokens may be NULL);
 *   non-head stage : the rows are replaced by this stage's output (the next stage's input);
 *   head stage     : logits_out (host f32[n_vocab], may be NULL) receives the last token's logits.
 * A full-model stage may pass hidden_dev = NULL (then this IS thk_model_prefill).  Stream-ordered: the call blocks only for the token
 * upload of an embedding stage and for the logits read-back; thk_pp_send / thk_peer_send_bulk behind it are ordered on the stream. */
int thk_model_prefill_stage(thk_model* m, int32_t seq, const int32_t* tokens, float* hidden_dev, int32_t n_tokens, int32_t n_past,
                            float* logits_out);
/* The prefill GEMMs stream "tile images" of the layer matrices: a second copy of this stage's layer weights in HBM (12.4 GB for
 * 7B), built by the first thk_model_prefill call unless this call built it earlier - so that the time and the memory are paid
 * when the embedder chooses.  Returns THK_ERR_OOM (thk_last_error explains) when the copy does not fit; prefill then still works
 * on the row-major matrices, ~20 % slower.  A weight write or a prefill_tile_* change makes the next call (or prefill) rebuild. */
int thk_model_prepare_prefill(thk_model* m);
/* 1 when the next thk_model_prefill will stream tile images, 0 when it will stream the row-major matrices. */
int thk_model_prefill_uses_tile_images(const thk_model* m);

/* Stream-ordered decode loop (no host round trip per token; what bench.py times).
 * Device-resident per-sequence state: position, current token, generated-token log. */
int thk_model_seq_set(thk_model* m, int32_t seq, int32_t token, int32_t pos);      /* async; clears the token log */
/* Override only the next input token (prompt tokens fed through the decode loop). */
int thk_model_seq_set_token(thk_model* m, int32_t seq, int32_t token);
/* Enqueue one decode step of sequence `seq` on the stream (graph replay):
 *   embed stage: x = emb[token[seq]]  else x = *hidden_in (dev f32[E])
 *   head stage : logits -> greedy token; token[seq] = it; appended to the log
 *   else       : *hidden_out (dev f32[E]) = stage output
 *   advance != 0: pos[seq] += 1 afterwards (0 = keep re-evaluating the same slot,
 *                 the fixed-T=512 benchmark protocol of BASELINE.md) */
int thk_model_decode_step(thk_model* m, int32_t seq, int advance);
/* n_steps back-to-back decode steps; replays captured multi-step graphs - one graph of exactly n_steps steps up to 32
 * (20 steps = ONE graph launch), 32-step graphs plus one remainder graph beyond - because consecutive graph launches sit
 * ~50 us apart on the GPU (measured on MI355X: single-step replays run 58 us per step slower than multi-step graphs).
 * Position contract (both calls): a step at position p needs p < n_ctx; an advancing step leaves p + 1.
 * A call that would evaluate a position >= n_ctx returns THK_ERR_INVALID and enqueues nothing (the device
 * side additionally never advances past n_ctx - 1). */
int thk_model_decode_steps(thk_model* m, int32_t seq, int32_t n_steps, int advance);
/* Capture now (nothing is run) the multi-step graphs thk_model_decode_steps(n_steps) will replay, so the first
 * timed call does not pay for stream capture + graph instantiation. */
int thk_model_prepare_steps(thk_model* m, int32_t seq, int32_t n_steps);
/* 1 when the finalized model runs a decode step as ONE persistent loader/consumer launch (thk_engine.hip;
 * tunable "engine" = 1, default 0, shape permitting), 0 when it runs 5 fused launches per layer. */
int thk_model_uses_engine(const thk_model* m);
/* Development aid: copy out one of the working buffers the last decode step left behind ("x" final hidden state, "q", "u",
 * "part_o", "part_ml": the LAST layer's), to compare two launch configurations stage by stage on a one-layer model. */
int thk_model_debug_buffer(thk_model* m, const char* name, float* out, int64_t cap, int64_t* n_out);
/* Development aid (tunable engine_trace=1 before finalize): the last step's per-workgroup, per-op s_memtime stamps,
 * [n_cu][n_ops][8] 64-bit words (slot meaning in thk_engine.hip). */
int thk_model_engine_trace(thk_model* m, unsigned long long* out, int64_t cap_words, int32_t* n_cu, int32_t* n_ops);
void* thk_model_hidden_in(thk_model* m, int32_t seq);   /* dev f32[E], RCCL recv target */
void* thk_model_hidden_out(thk_model* m, int32_t seq);  /* dev f32[E], RCCL send source */
void* thk_model_token_dev(thk_model* m, int32_t seq);   /* dev int32: current/next token id */
void* thk_model_logits_dev(thk_model* m, int32_t seq);  /* dev f32[V] (head stage) */
/* Blocking: copy the generated-token log (up to cap ids) and position of `seq`. */
int thk_model_seq_get(thk_model* m, int32_t seq, int32_t* tokens_out, int32_t cap, int32_t* n_out, int32_t* pos_out);
/* Logits of the sequence's last evaluated token (head stages): the k largest via thk_topk_f32's kernel, or all n_vocab of them. */
int thk_model_logits_topk(thk_model* m, int32_t seq, int32_t k, float* values_out, int32_t* ids_out);
/* thk_model_eval and thk_model_logits_topk in ONE stream round trip (what a stochastic sampler needs per token, th-llama.cpp:686-724 without
 * the 4 * n_vocab-byte mapping): the steps, the selection kernel behind them, its k keys written into host-mapped memory, one synchronisation. */
int thk_model_eval_topk(thk_model* m, int32_t seq, const int32_t* tokens, int32_t n_tokens, int32_t n_past, int32_t k, float* values_out, int32_t* ids_out);
int thk_model_read_logits(thk_model* m, int32_t seq, float* logits_out);
/* Device-side step clock (head stages): clock_out[i] = the chip-wide 100 MHz counter (s_memrealtime) at the end of the step that
 * logged token i of thk_model_seq_get.  Differences are per-step durations taken on the GPU, inside replayed multi-step graphs,
 * with nothing added to the stream.  No reference counterpart (the reference times whole passes on the host, th-llama.cpp:640-655). */
int thk_model_seq_clock(thk_model* m, int32_t seq, unsigned long long* clock_out, int32_t cap, int32_t* n_out);
/* The sequence's current token: the greedy pick of its last step (or what thk_model_seq_set / _set_token put there).  One
 * 4-byte read-back after a stream sync; unlike the log of thk_model_seq_get it does not depend on how many steps ran. */
int thk_model_seq_last_token(thk_model* m, int32_t seq, int32_t* token_out);

/* Bytes of HBM this stage streams per decode step at context length T
 * (weights + KV read + KV write + gains; SURVEY.md §8d formula) — used by bench.py. */
int64_t thk_model_bytes_per_token(const thk_model* m, int32_t T);
/* Diagnostics for bench.py: runs ONE eager (un-graphed) hold-position decode step of sequence `seq` with a HIP event between the
 * launches and returns, per launch in issue order, its name and the milliseconds to the next event (so each figure carries the
 * dispatch gap a graph replay does not pay).  The sequence does not advance (its advance setting is restored); at most
 * max_entries entries are written, *n_out says how many.  No reference counterpart (the reference times whole passes). */
int thk_model_profile_step(thk_model* m, int32_t seq, int32_t max_entries, char (*names)[48], float* ms, int32_t* n_out);
/* Development aid; needs a library built with -DTHK_TRACE (libthk_trace.so), THK_ERR_STATE otherwise.  Runs TWO hold-position
 * decode steps as one replayed graph (ONE eager step with use_graph = 0) in which every wave of every launch stamps the 100 MHz
 * s_memrealtime counter at four points: kernel entry | activation vector staged | first weight batch consumed | done; the stamps
 * of the LAST step are returned.  The sequence does not advance.  out = [n_kernels][blocks_per_kernel][8 waves][4] u64, 0 = not
 * stamped; names in thk_model_profile_step order.  No reference counterpart (the reference times whole passes, th-llama.cpp:640-655). */
int thk_model_step_trace(thk_model* m, int32_t seq, unsigned long long* out, int64_t cap_words, int32_t max_names, char (*names)[48],
                         int32_t* n_kernels, int32_t* blocks_per_kernel);

/* ---------------------------------------------------------------- pipeline hand-off (config C4)
 * Point-to-point RCCL over xGMI between pipeline stages (one process per GPU).  New functionality:
 * the reference is single-device.  The 128-byte id comes from thk_pp_get_unique_id
 * on rank 0 and is distributed by the host out of band (file, socket, torch.distributed ...).
 * All transfers are enqueued on the context's stream; wrap a ring step (one send + one receive) in
 * thk_pp_group_begin/end so it cannot dead-lock. */
typedef struct thk_pp thk_pp;
#define THK_PP_UNIQUE_ID_BYTES 128
int thk_pp_get_unique_id(void* out128);
int thk_pp_create(thk_ctx* ctx, int n_ranks, int rank, const void* unique_id128, thk_pp** out);
int thk_pp_destroy(thk_pp* pp);
int thk_pp_rank(const thk_pp* pp);
int thk_pp_size(const thk_pp* pp);
int thk_pp_group_begin(thk_pp* pp);
int thk_pp_group_end(thk_pp* pp);
int thk_pp_send(thk_pp* pp, const void* dev_buf, size_t bytes, int peer);
int thk_pp_recv(thk_pp* pp, void* dev_buf, size_t bytes, int peer);
int thk_pp_send_hidden(thk_pp* pp, thk_model* m, int32_t seq, int peer);   /* thk_model_hidden_out(m,seq), n_embd*4 bytes */
int thk_pp_recv_hidden(thk_pp* pp, thk_model* m, int32_t seq, int peer);   /* into thk_model_hidden_in(m,seq) */
int thk_pp_send_token(thk_pp* pp, thk_model* m, int32_t seq, int peer);    /* last stage -> stage 0: 4-byte greedy token */
int thk_pp_recv_token(thk_pp* pp, thk_model* m, int32_t seq, int peer);

/* ---- the same hand-off without a communication library (token-hawk_amd/csrc/thk_peer.hip; SURVEY.md 8(e) fallback): each
 * stage owns a mailbox in its HBM, exported with hipIpcGetMemHandle (dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0); the producing
 * stage's stream stores the hidden state (n_embd*4 bytes) or the 4-byte token straight into the NEXT stage's mailbox and
 * raises a sequence-numbered flag, the consuming stage's stream waits for the flag (bounded, ~2 s) and copies the payload into
 * thk_model_hidden_in / thk_model_token_dev.  Opt-in (bench.py --transport peer).  Usage per stage: create -> export the
 * 64-byte handle -> hand it to the PREVIOUS stage by any means -> connect(handle of the NEXT stage; NULL = single-stage ring)
 * -> per micro-step send(...) then recv(...) on the context's stream -> check for time-outs at sync points.
 * No reference counterpart (the reference is single-device). */
typedef struct thk_peer thk_peer;
#define THK_PEER_HANDLE_BYTES 64
enum { THK_PEER_HIDDEN = 0, THK_PEER_TOKEN = 1, THK_PEER_BULK = 2 /* thk_peer_send_bulk / _recv_bulk only */ };
/* how the mailbox was allocated: uncached / fine-grained device memory is visible to a polling kernel while ANOTHER GPU writes it;
 * coarse-grained (plain hipMalloc: the fallback when the runtime cannot allocate or IPC-export the others) only guarantees that
 * at dispatch boundaries, i.e. it is safe for rings inside one GPU only */
enum { THK_PEER_MEM_COARSE = 0, THK_PEER_MEM_UNCACHED = 1, THK_PEER_MEM_FINEGRAINED = 2 };
int thk_peer_create(thk_ctx* ctx, thk_model* stage, int32_t n_seq, thk_peer** out);
int thk_peer_export(thk_peer* p, void* handle_out64);
int thk_peer_connect(thk_peer* p, const void* next_handle64);
int thk_peer_send(thk_peer* p, int32_t seq, int kind);   /* kind: THK_PEER_HIDDEN (thk_model_hidden_out) | THK_PEER_TOKEN (thk_model_token_dev) */
int thk_peer_recv(thk_peer* p, int32_t seq, int kind);   /* into thk_model_hidden_in | thk_model_token_dev */
/* Bulk hand-off (round 6): `bytes` of a caller-owned device buffer - the f32 [n_tokens, n_embd] rows thk_model_prefill_stage leaves for the
 * next stage - through the sequence's bulk slot of the mailbox (capacity n_ctx * n_embd * 4 bytes per sequence; bytes % 16 == 0).  A slot
 * holds ONE payload: send the same sequence's next prompt only after the consumer has taken this one (synchronise + fence across ranks).
 * A receive whose wait times out (~2 s) copies nothing and raises the sticky error word thk_peer_check reports. */
int thk_peer_send_bulk(thk_peer* p, int32_t seq, const void* src_dev, size_t bytes);
int thk_peer_recv_bulk(thk_peer* p, int32_t seq, void* dst_dev, size_t bytes);
int thk_peer_check(thk_peer* p);                         /* THK_ERR_STATE once a wait has timed out: sticky - send / recv refuse from then on, destroy and recreate */
int thk_peer_memory_kind(const thk_peer* p);             /* THK_PEER_MEM_* of this stage's mailbox */
int thk_peer_destroy(thk_peer* p);

/* Tuning knobs (integers, by name) so the bench can sweep launch geometry without
 * rebuilding.  Decode knobs must be set before thk_model_finalize; prefill knobs are read
 * per call.  Unknown names return THK_ERR_NOTFOUND.
 *   decode : gemv_blocks_per_cu; gemv_bpc_{qkv,wo,w13,w2,head} and gemv_variant_{...}
 *            (-1 = per-shape default; 0, 1, 2 batch loops - row pairs, single rows, row pairs in half batches -, 5, 6
 *            software-pipelined loops - row pairs, single rows; for qkv and w13 single rows meet their RoPE / SwiGLU partner
 *            in LDS -, w2 only: 8 = a workgroup per row, a wave per quarter of it; 3, 4, 7, 9 are retired numbers);
 *            gemv_grid_{...} (> 0: that many workgroups, whatever gemv_bpc_* says); attn_splits (0 = auto: 4, or 8 for caches longer than 512 rows; 1|2|4|8);
 *            attn_waves (0 = auto: 8 waves per attention workgroup, 16 for f32 caches longer than 1024 rows; 4|8|16); use_graph; kv_f16 (1 = K/V caches stored as binary16, rounded RNE at the append: half the
 *            KV bytes, thk_model_bytes_per_token then counts s_kv = 2; default 0 = f32 like the reference,
 *            th-llama-loader.cpp:335); engine (1 = persistent loader/consumer launch per step when the
 *            shape allows, 0 = launches); attn_tc_dyn, fold_embed; fold_finish (1, default: ONE workgroup
 *            of the lm-head launch - the highest-numbered - polls the other workgroups' arg-max key slots and finishes the token inside
 *            the launch.  Forward progress does not depend on the dispatch order: the poller holds one workgroup slot, all others keep
 *            taking the launch's workgroups; the order only decides how long it spins.  2 = workgroup 0 polls instead (dispatched
 *            first: the adversarial placement, kept as a tested mode).  The wait is bounded by a 1 s device time-out - only a GPU
 *            whose CUs are held by OTHER processes for that long can trip it (bench.py --ranks-share-gpu therefore runs with 0) ->
 *            SeqState error word -> THK_ERR_STATE from thk_model_seq_get / seq_last_token, which also clear the key slots;
 *            0 = the pick as a launch of its own, no in-launch wait, kept tested);
 *            measure_skip_kernel (1..6: that kernel is not launched -- bench.py's marginal-cost
 *            measurement; results are garbage; REFUSED unless the environment has THK_MEASURE_HOOKS=1)
 *   prefill: prefill_slab_tokens (256, default: up to 256 prompt tokens share one pass over the weights; 128 = the
 *            four-token-tile kernels only); prefill_deferred_norm (1, default: RMSNorm's per-token scalar is applied on
 *            the output side of the GEMM, two launches per layer fewer; 0 = norm -> image launches);
 *            prefill_blocks_{qkv,wo,w13,w2} (workgroups per GEMM launch, <= 256; 0 = auto);
 *            prefill_tile_{...} (weight rows per workgroup, 128|256); prefill_attn_mfma;
 *            prefill_packed (default 1: the first prefill call makes tile images of the
 *            layer matrices for the GEMM's linear `nt` stream — a second copy of the layer
 *            weights in HBM, rebuilt after any tensor write; 0 = stream the row-major matrices) */
int thk_set_tunable(thk_ctx* ctx, const char* name, int64_t value);
int thk_get_tunable(thk_ctx* ctx, const char* name, int64_t* value);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* THK_H */
