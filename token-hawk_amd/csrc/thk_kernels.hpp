// thk_kernels.hpp — launch interface between the C-ABI layer (thk_ctx.cpp, thk_ops.cpp, thk_model.cpp) and the
// HIP kernels (thk_kernels.hip, thk_prefill.hip).  Internal; not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace thk {

constexpr int kBlock = 256;   // threads per workgroup (4 waves of 64)
constexpr int kWaves = 4;
constexpr int kMaxSplit = 8; // attention context splits per head

// Device-resident per-sequence decode state (read by kernels so one hipGraph serves all tokens).
struct SeqState {
    int32_t pos;     // n_past of the token being evaluated (T = pos + 1)
    int32_t token;   // current input token id / greedy result of the last step
    int32_t n_gen;   // tokens appended to the log so far
    int32_t pad;     // error word: 1 = the folded greedy pick gave up waiting for an arg-max key (reported by thk_model_seq_get)
};

// prologue / epilogue selectors of the fused mat-vec
enum { GEMV_PRO_COPY = 0, GEMV_PRO_RMS = 1, GEMV_PRO_ATTN = 2, GEMV_PRO_RMS_EMBED = 3 };
enum { GEMV_EPI_STORE = 0, GEMV_EPI_RESID = 1, GEMV_EPI_ROPE_KV = 2, GEMV_EPI_SWIGLU = 3, GEMV_EPI_HEAD = 4 };

// the greedy pick + sequence bookkeeping that ends a decode step (finish_token_reduce, thk_decode_bodies.hpp): arguments of the
// stand-alone launch, and - with `folded` set - of the lm-head launch whose highest-numbered workgroup does the same work itself
struct FinishArgs {
    const unsigned long long* block_best; int nblocks;     // stand-alone launch only (the lm-head launch reads its own GemvArgs::block_best)
    SeqState* st; int32_t* gen_log; int log_cap;
    const int* advance_ptr; int32_t* id_out; int n_ctx; unsigned* epoch;
    unsigned long long* trace; unsigned long long* clock_log;
    int folded;                // lm-head launch only: 1 = this launch finishes the token itself (its key slots must be zero when it starts)
};

struct GemvArgs {
    const uint16_t* W[3];   // f16 row-major [R,C] matrices (see gemv_kernel for their meaning per epilogue)
    int R;                  // rows of W[0] (STORE/RESID/HEAD)
    int C;                  // columns (input features), multiple of 256
    int n_groups;           // row groups to process
    const float* x;         // PRO_COPY / PRO_RMS input vector f32[C]
    const float* gain;      // PRO_RMS gain f32[C]
    // PRO_RMS_EMBED: x = f32(embed[*tok_ptr, :]) (f16 table [n_vocab, C]); block 0 also stores it to x_out
    const uint16_t* embed; const int32_t* tok_ptr; float* x_out;
    float* y;               // output (q for ROPE_KV, u for SWIGLU, logits for HEAD)
    const float* resid;     // EPI_RESID
    // EPI_ROPE_KV
    float* kcache; float* vcache; const float* rope_tab; const int32_t* pos_ptr; int pos_val; int E; int D;
    int kv_f16;             // caches hold binary16 [n_ctx, H, D] (optional; the reference's and the default are f32)
    // PRO_ATTN
    const float* part_o; const float* part_ml; int H; int nsplit;
    unsigned d_magic;            // PRO_ATTN: ceil(2^32 / D), set by launch_gemv: element e -> head __umulhi(e, d_magic) (exact for e, D < 65536) instead of a
                                 // division between the prologue's requests.  Travels to the kernel, with D, in the unused W1 slot of the preloaded scalars.
    // EPI_HEAD
    int lm_faithful; int q1_split; int q1_cov; unsigned long long* block_best;
    FinishArgs fin;              // EPI_HEAD: fin.folded folds the greedy pick into this launch
    unsigned long long* trace;   // development timeline, [blocks][8 waves][4] (only read by a THK_TRACE build; NULL otherwise)
};

struct AttnArgs {
    const float* q;            // [H*D]
    const float* kcache;       // [n_ctx, H, D] f32
    const float* vcache;
    const int32_t* pos_ptr; int pos_val;   // T = pos + 1
    int H, D, nsplit, tc;      // tc = positions per split ...
    int tc_dyn;                // ... or 1: tc = ceil(T / nsplit) rounded up to the wave batch, computed on the device from the live position
    unsigned ns_magic;         // ceil(2^32 / nsplit), set by launch_attn_decode: x / nsplit = __umulhi(x, ns_magic) for x * nsplit < 2^32 (no integer
                               // division between the kernel's entry and its K/V requests); reaches the kernel in the preloaded scalars
    int pipe;                  // 1: a split may span several rounds of the workgroup (long caches): take the software-pipelined variant where it exists
    int waves;                 // waves per block of the stand-alone kernel (4 or 8)
    int nq;                    // causal queries in this launch (prefill); 0/1 = single decode query
    int kv_f16;                // the caches hold binary16 (kcache / vcache then point to _Float16 data)
    float scale;               // 1/sqrtf(D)
    float* out;                // finished output [H*D] (nsplit == 1); NULL = write split partials
    float* part_o;             // [H, nsplit, D]
    float* part_ml;            // [H, nsplit, 2]
    unsigned long long* trace; // development timeline, [blocks][8 waves][4] (only read by a THK_TRACE build; NULL otherwise)
};

// nru (0..7) selects the (rows per wave iteration, slots per load batch) variant for the column class of C;
// see gemv_variant() in thk_kernels.hip.  Row groups = ceil(rows / gemv_rows_per_group).
int gemv_rows_per_group(int C, int epi, int nru);
void gemv_variant(int C, int epi, int nru, int* NR, int* U, int* pipe);
hipError_t launch_gemv(int pro, int epi, int nru, const GemvArgs& a, int grid, bool nt, hipStream_t st);
// y = resid + W x with a WORKGROUP per row (its four waves take a quarter of the columns each; round 4, thk_decode_bodies.hpp):
// gemv_quarter_ok = the shape and workgroup count have this form (<= 32 rows per workgroup)
bool gemv_quarter_ok(int C, int R, int grid);
hipError_t launch_gemv_quarter(const GemvArgs& a, int grid, hipStream_t st);
hipError_t launch_attn_decode(const AttnArgs& a, hipStream_t st);
// positions one round of an attention workgroup covers (waves x positions per wave-instruction x wave-instructions per batch): a split longer than this takes the pipelined variant
int attn_round_positions(int D, int waves, bool kv_f16);
hipError_t launch_attn_combine(const float* part_o, const float* part_ml, float* out, int H, int D, int nsplit, hipStream_t st);
hipError_t launch_rms_norm(float* x, int rows, int N, hipStream_t st);
hipError_t launch_row_mul(float* x, const float* g, int rows, int N, hipStream_t st);
hipError_t launch_rope(float* x, const float* tab, int n_tok, int H, int D, int n_past, hipStream_t st);
hipError_t launch_row_softmax(float* x, int rows, int N, hipStream_t st);
hipError_t launch_add(const float* a, const float* b, float* c, size_t n, hipStream_t st);
hipError_t launch_silu(float* a, size_t n, hipStream_t st);
hipError_t launch_mul(float* a, const float* b, size_t n, hipStream_t st);
hipError_t launch_kv_append(float* kc, float* vc, const float* k, const float* v, int pos, int E, hipStream_t st);
hipError_t launch_embed_rows(const uint16_t* table, const int32_t* tokens_dev, int n, int E, float* x, hipStream_t st);
hipError_t launch_embed(const uint16_t* table, const SeqState* st_dev, int token_val, int E, float* x, hipStream_t st, unsigned long long* trace = nullptr);
hipError_t launch_finish_token(const unsigned long long* block_best, int nblocks, SeqState* st_dev, int32_t* gen_log, int log_cap,
                               const int* advance_ptr, int32_t* id_out, int n_ctx, unsigned* epoch, hipStream_t st, unsigned long long* trace = nullptr,
                               unsigned long long* clock_log = nullptr);   // clock_log[i] = s_memrealtime (100 MHz) when the step that logged token i finished
hipError_t launch_finish_token_args(const FinishArgs& a, hipStream_t st);
constexpr int kTraceBlocks = 2048;   // workgroups recorded per launch of the development timeline
constexpr int kTraceWords = 8 * 4;    // u64 per workgroup: [wave (8)][stamp (4)]
bool trace_compiled();               // true in a -DTHK_TRACE build (libthk_trace.so)
hipError_t launch_advance_pos(SeqState* st_dev, const int* advance_ptr, int n_ctx, unsigned* epoch, hipStream_t st);
hipError_t launch_argmax(const float* logits, int V, unsigned long long* block_best, int nblocks, hipStream_t st);
// the k largest logits as sorted keys ((ordered value << 32) | ~index, descending); V <= 32768, k <= 1024
// done != NULL: keys_out (and done) are host-mapped; the kernel stores `epoch` there, system scope, after its keys (a host thread polls it)
hipError_t launch_topk(const float* logits, int V, int k, unsigned long long* keys_out, unsigned long long* cand_scratch /* topk_scratch_bytes(V, k) device bytes */, hipStream_t st,
                       unsigned long long* done = nullptr, unsigned long long epoch = 0);
size_t topk_scratch_bytes(int V, int k);
hipError_t launch_synth_f16(uint64_t key, float scale, size_t n, void* out, hipStream_t st);
hipError_t launch_synth_gain(uint64_t key, float scale, size_t n, float* out, hipStream_t st);

// ---------------------------------------------------------------- thk_engine.hip: persistent loader/consumer decode engine
constexpr int kMaxDevices = 64;
constexpr int kEngSlotBytes = 16384;      // one ring slot = one fill = up to 16 pieces of 1 KiB (64 lanes x 16 B)
enum { EOP_QKV = 0, EOP_ATTN = 1, EOP_WO = 2, EOP_W13 = 3, EOP_W2 = 4, EOP_HEAD = 5 };
enum { EIN_GRAN = 0, EIN_PLAIN = 1 };
// One op of the engine's program (device array, immutable after finalize).  A "unit" is what one consumer wave turns
// into finished outputs: two weight rows (a row pair of one matrix, or row g of w1 and row g of w3 when dual).
struct EngOp {
    int kind;                 // EOP_*
    int n_units;              // units of the op; CU c takes units c, c + n_cu, ...
    int fpu;                  // fills (ring slots) per unit
    int pieces_last;          // 1 KiB pieces in the unit's last fill (the others carry 16)
    int C;                    // input features (length of the activation vector)
    int row_bytes;            // C * 2
    int dual;                 // unit = row u of W[0] followed by row u of W[1]
    int in_src;               // EIN_GRAN: 8-byte {value, tag} granules published by op in_tag_op; EIN_PLAIN: f32 written before the launch
    const uint16_t* W[3];
    const float* gain;        // non-null: RMSNorm * gain applied while the input is gathered
    const void* in_ptr;
    int in_n;                 // elements of the input vector (multiple of 64, >= 192)
    int in_tag_op;
    int in_dst;               // LDS vector 0 (large) or 1
    int resid_src;            // 0 none, 1 granules, 2 plain f32 (element 2u, 2u+1 for unit u)
    const void* resid_ptr;
    unsigned long long* out_g;   // granules out (QKV: [3E] q | k_new | v_new; ATTN: attention output [E])
    float* out_plain;         // optional f32 copy of the outputs (hidden state for the host / next stage, logits)
    float* kcache; float* vcache;                 // QKV (append) and ATTN (read): this layer's f32 [n_ctx, H, D] caches
    const unsigned long long* qg;                 // ATTN: the QKV op's granules
    unsigned long long* pg;                       // ATTN: split partials [H, nsplit, D + 2]
};
struct EngArgs {
    const EngOp* ops; int n_ops;
    const SeqState* st; const unsigned* epoch; unsigned* err;
    int E, H, D, nsplit, tc;
    int NS, v0_bytes, v1_bytes;                   // LDS plan: ring slots, bytes of activation vectors 0 and 1
    const float* rope_tab; float scale;
    unsigned long long* block_best;               // HEAD: one arg-max key per workgroup
    unsigned long long* trace;                    // development timeline (NULL in production), see thk_engine.hip
};
size_t engine_lds_bytes(int NS, int v0_bytes, int v1_bytes);
hipError_t launch_engine(const EngArgs& a, int n_cu, hipStream_t st);

// Byte offset of a kernel's LAST parameter in its kernarg segment = size of the explicit leading scalars in front of the by-value argument struct.
// The build preloads exactly 14 dwords (hipcc -mllvm -amdgpu-kernarg-preload-count=14, __graft_entry__.py) and the kernels that depend on it read
// those scalars before their first s_waitcnt: adding, removing or reordering a leading argument must be a compile error, not a silent slow path.
template <class F> struct KernargLead;
template <class... A> struct KernargLead<void (*)(A...)> {
    static constexpr size_t bytes() {
        constexpr size_t n = sizeof...(A);
        const size_t sz[] = {sizeof(A)...}, al[] = {alignof(A)...};
        size_t off = 0;
        for (size_t i = 0; i + 1 < n; ++i) off = (off + al[i] - 1) / al[i] * al[i] + sz[i];
        return (off + al[n - 1] - 1) / al[n - 1] * al[n - 1];
    }
};
constexpr size_t kKernargPreloadBytes = 14 * 4;

// thk_prefill.hip
hipError_t launch_gemm_f16_prefill(const uint16_t* W, int R, int C, const float* X, int M, float* Y, void* workspace, hipStream_t st);
size_t gemm_prefill_workspace_bytes(int M, int R, int C);
// v3 (LDS-DMA pipeline + stream-K): activations are staged once as an "X image" (hi/lo f16, the LDS layout of
// the MFMA loop), up to 3 matrices sharing the operand run in one launch, partial tiles are reduced by a
// kernel that carries the fused epilogue.  M <= 128 tokens per call.
struct PrefillPlan {
    int M, MT, Mpad, R, Rpad, nmat, C, nchunks, rb_per_mat, rb_total, per, maxseg, G, tile_rows;
    int packed;         // 1: W[] point at tile images made by launch_prefill_pack with this plan's tile_rows (linear `nt` stream), 0: row-major matrices
    size_t ximg_bytes, slot_floats, part_floats;
};
PrefillPlan prefill_plan(int M, int R, int nmat, int C, int G /* workgroups; 0 = default (256) */, int tile_rows /* 128 | 256 (default) */);
// ssq != NULL (with a gain): DEFERRED norm - the image holds x * 2^k * gain (2^k = the power of two below 1/rms), the row's sum of squares is added
// to ssq[token] (2^-32 fixed point, must be zero before), and the reducer of the GEMM that consumes the image applies 1/rms / 2^k
// (launch_prefill_reduce_qkv / _swiglu with ssq = ssq_scale = this array)
hipError_t launch_prefill_ximg(const float* X, const float* gain_or_null, int M, int C, void* ximg, hipStream_t st, unsigned long long* ssq = nullptr);
// X += sum of the partial tiles (residual add), and the deferred-norm image of the result for the next GEMM (x * gain, sum of squares -> ssq)
// (2^k from ssq_scale: the sum of squares of the token's PREVIOUS norm input - the consumer is given the same array)
hipError_t launch_prefill_reduce_resid_ximg(const float* part, const PrefillPlan& p, float* X, const float* gain, void* ximg, unsigned long long* ssq,
                                            const unsigned long long* ssq_scale, hipStream_t st);
hipError_t launch_prefill_gemm(const uint16_t* const* W, const PrefillPlan& p, const void* ximg, float* part, hipStream_t st);
size_t prefill_pack_bytes(int R, int C, int tile_rows);
hipError_t launch_prefill_pack(const uint16_t* W, int R, int C, int tile_rows, void* out, hipStream_t st);
hipError_t launch_prefill_reduce_store(const float* part, const PrefillPlan& p, float* Y, bool residual, hipStream_t st);
hipError_t launch_prefill_reduce_qkv(const float* part, const PrefillPlan& p, const float* rope_tab, int n_past, int D, float* Q, void* kcache, void* vcache, bool kv_f16, hipStream_t st,
                                     const unsigned long long* ssq = nullptr, const unsigned long long* ssq_scale = nullptr);
// causal attention for the M queries of a slab (D = 64 | 128), f32-class accuracy on MFMA
hipError_t launch_attn_prefill_mfma(const float* Q, const void* Kc, const void* Vc, bool kv_f16, int n_past, int M, int H, int D, float* out /* [M,H*D] or null */,
                                    void* ximg /* if out is null: the X image of the consuming GEMM */, hipStream_t st,
                                    int img_MT = 0 /* token tiles of that image (0: those of M) */, int img_tok0 = 0 /* image row of query 0 */,
                                    int q_tiles = 0 /* > the queries' own tiles: the extra 32-query tiles are written as zero rows */);
int prefill_token_tiles(int M);     // token tiles of a slab of M tokens: 1..4 up to 128 tokens, 8 for 129..256
hipError_t launch_prefill_reduce_swiglu(const float* part, const PrefillPlan& p, void* ximg_out, hipStream_t st, const unsigned long long* ssq = nullptr,
                                        const unsigned long long* ssq_scale = nullptr);

}  // namespace thk
