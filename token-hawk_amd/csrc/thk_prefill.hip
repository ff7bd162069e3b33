// thk_prefill.hip — batched prompt-prefill GEMM on the CDNA4 matrix cores (config C3).
//
//   Y[M,R] = X[M,C] * W[R,C]^T     X,Y f32 row-major, W GGML f16 row-major.
//
// This is the one place on the path where the work is a real dense contraction
// (M = 128 prompt tokens share every weight row), so it goes to MFMA
// (v_mfma_f32_32x32x16_f16) instead of the HBM-streaming mat-vec.  The reference
// multiplies f32 activations by f16-decoded weights in f32 (cmdbuf_mat_mul with f16 B,
// th.cpp:396-539, batch branch th-llama.cpp:307-311); to keep that precision on f16
// matrix cores the activations are split x = hi + lo (both f16, |lo| <= 2^-11 |x|) and
// two MFMAs accumulate into the same f32 tile: error ~2^-22 relative, f32-class.
//
// Fragment layout used (gfx950 32x32x16, 8 f16 per lane per operand):
//   A[i = lane&31][k = 8*(lane>>5) + e], B[k = 8*(lane>>5) + e][j = lane&31], e = 0..7
//   D[row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = lane&31], reg = 0..15
// We set A = 32 weight rows, B = 32 tokens, so both operands are 16-byte contiguous
// loads along K straight from row-major memory (no transposes), and each lane ends up
// with 4 consecutive output features of one token => float4 stores.
//
// Structure: see the block comment above gemm_prefill_v3_kernel (LDS-DMA pipeline, stream-K, fused reducers).
// Two earlier kernels (one wave per workgroup; 4 waves with register-staged X and split-K) ran at the memory
// latency -- 46 us per launch, 2.4x fetch amplification -- and were removed; profiles/ keeps their numbers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "thk_kernels.hpp"
#include <type_traits>

namespace thk {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum_prefill(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Diagnostic builds (tools/dev/gemm_diag.sh; results are WRONG, only the timing is of interest - DESIGN.md 4.3 "what bounds the GEMM loop"):
//   -DTHK_PF_FAKEW  every workgroup streams the same eight weight tiles (L2-resident): the loop with the HBM stream taken out
//   -DTHK_PF_NOLO   the lo-piece MFMAs are skipped: the loop with half the matrix work
#ifdef THK_PF_FAKEW
#define THK_WTILE(RBL, CH) ((size_t)((CH) & 7))
#else
#define THK_WTILE(RBL, CH) ((size_t)(RBL) * nchunks + (CH))
#endif
//   -DTHK_PF_NOREAD the 128-token loop never re-reads its fragments from LDS (it multiplies the first chunk's over and over): the loop without its ds_reads
//   -DTHK_PF_NODMA  the 128-token loop issues no LDS-DMA behind the prologue's: the loop without its fill
#ifdef THK_PF_NOLO
constexpr bool kNoLo = true;
#else
constexpr bool kNoLo = false;
#endif
#ifdef THK_PF_NOREAD
constexpr bool kNoRead = true;
#else
constexpr bool kNoRead = false;
#endif
#ifdef THK_PF_NODMA
constexpr bool kNoDma = true;
#else
constexpr bool kNoDma = false;
#endif
constexpr int kKC = 32;                  // K columns per LDS stage (2 MFMA k-steps)

static int current_device() { int d = 0; (void)hipGetDevice(&d); return d >= 0 && d < kMaxDevices ? d : 0; }

// ---------------------------------------------------------------- LDS-DMA pipeline, stream-K
// A kernel that keeps one K-chunk of loads in flight per wave runs, at one workgroup per CU, at the memory
// LATENCY (measured: 46 us per launch, 2.4 TB/s of fetch for 1.3 TB/s of weights).  This one is built
// around memory-level parallelism instead:
//   * activations are written ONCE per operand into the exact LDS image the MFMA loop reads
//     ("X image": per 32-column K-chunk, [hi|lo][token][64 B], XOR-swizzled pieces => conflict-free
//     ds_read_b128), so a stage is a linear 1 KiB-per-instruction global_load_lds copy;
//   * weights also arrive by global_load_lds (quad-coalesced, same swizzle), so no VGPR is tied up
//     by data in flight and hipcc's waitcnt bookkeeping is out of the picture: completion is counted
//     by hand (s_waitcnt vmcnt(N), loads retire in order);
//   * kNST LDS stages of 32 KB => kNST chunks = 128 KB per CU in flight; one raw s_barrier per chunk; the loop is software-
//     pipelined across chunks (fragments of chunk g+1 are read and the DMA of chunk g+kNST is issued BETWEEN the MFMAs of
//     chunk g, see the kernel) because one wave per SIMD hides nothing by occupancy;
//   * the layer matrices are streamed from TILE IMAGES (pack_w_kernel: a second copy of the weights in exactly the LDS stage
//     layout, made on the first prefill call) with `nt`, or from the row-major matrices when that copy does not fit;
//   * a wave owns 64 weight rows (two A fragments share every B fragment read), a workgroup 256 rows x 128 tokens
//     (NF = 1: 32 rows per wave, 128-row tiles -- half the partial-tile bytes for matrices with few row-blocks);
//   * stream-K: the (row-block, K-chunk) space of up to three matrices that share the operand
//     (wq|wk|wv, w1|w3) is flattened and cut into kG equal contiguous shares, one per workgroup, so
//     every CU streams the same number of bytes whatever the shape.  A workgroup's share spans at
//     most `maxseg` row-blocks; each span goes to its own partial slot and the reducers below sum
//     the slots of a row-block in workgroup order (deterministic, no atomics) and apply the fused
//     epilogue (store | residual add | RoPE + KV-cache write | SwiGLU + hi/lo image for w2).
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
constexpr int kV3Rows = 256;       // default weight rows per workgroup (4 waves x 64); 128 (4 x 32) halves the partial-tile bytes
constexpr int kNST = 4;            // LDS stages
constexpr int kG = 256;            // default workgroups per launch: a constant, so results do not depend on the CU count

static __host__ __device__ inline size_t ximg_stage_bytes(int MT) { return (size_t)MT * 32 * 64 * 2; }
// token tiles of a slab: 1 .. 4 for up to 128 tokens, EIGHT for 129 .. 256 (gemm_prefill_v3h_kernel: two halves of four tiles; pad tiles are zero rows)
int prefill_token_tiles(int M) { return M > 128 ? 8 : (M + 31) / 32; }

PrefillPlan prefill_plan(int M, int R, int nmat, int C, int G, int tile_rows) {
    PrefillPlan p{};
    if (G <= 0) G = kG;
    p.G = G;
    p.tile_rows = tile_rows == 128 ? 128 : kV3Rows;
    p.M = M; p.MT = prefill_token_tiles(M); p.Mpad = p.MT * 32; p.R = R; p.nmat = nmat; p.C = C;
    p.nchunks = C / kKC;
    p.rb_per_mat = (R + p.tile_rows - 1) / p.tile_rows; p.Rpad = p.rb_per_mat * p.tile_rows; p.rb_total = p.rb_per_mat * nmat;
    const long total = (long)p.rb_total * p.nchunks;
    p.per = (int)((total + G - 1) / G);
    p.maxseg = (p.per - 1 + p.nchunks - 1) / p.nchunks + 1;
    p.ximg_bytes = (size_t)p.nchunks * ximg_stage_bytes(p.MT);
    p.slot_floats = (size_t)p.Mpad * p.tile_rows;
    p.part_floats = (size_t)G * p.maxseg * p.slot_floats;
    return p;
}

// AUX = cache-policy bits (2 = nt).  Weights are NOT loaded nt: a 128-byte line is consumed as two 64-byte halves one
// chunk apart, and with nt the second half is fetched again (measured: +30 % FETCH_SIZE, 7.05 -> 7.7 ms).
template <int AUX = 0>
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, AUX);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Both operand tiles use one LDS layout: rows (tokens / weight rows) of 64 bytes = the chunk's 32 columns as four
// 16-byte pieces, piece p of row r stored at position p ^ ((r >> 2) & 3).  That XOR makes the MFMA fragment read
// (32 consecutive rows, one piece each) hit 16 different 16-byte bank groups per 16 lanes (SQ_LDS_BANK_CONFLICT = 0)
// without row padding, and it lets the LOADER run quad-coalesced: lanes 4r..4r+3 fetch the four pieces of row r,
// i.e. one 64-byte segment per quad, which the texture-address unit takes at 1 quad/clk.  The obvious
// "each lane fetches its own fragment" pattern (32 rows x 2x16 B per instruction) runs at 1 LANE per clock:
// 16 B/clk/CU instead of 55-64 (tools/probes/glds_rate_probe.hip, profiles/r01_glds_rate_probe.txt).
__device__ __forceinline__ int swz_pos(int row, int piece) { return piece ^ ((row >> 2) & 3); }

// The loads of one K-chunk into LDS stage `sb`, LPS = MT + 2 NF per wave: this wave's share of the X image (pieces wave, wave+4,
// ...: the image is already in the layout above, so a linear copy) and its own 32 NF weight rows (16 rows x 64 B per
// instruction).  Row-major weights: lane -> (row lane>>2 of each 16-row group, piece swz_pos(lane>>2, lane&3)), rows past the
// matrix repeat the last one (their outputs are never read).  Tile images (pack_w_kernel below): the W half of the stage is one
// contiguous TR x 64-byte block in global memory, a linear copy of full 128-byte lines that nobody else reads => `nt`.
// (Row-major weights make every instruction touch 16 rows x 64 bytes, and `nt` would fetch each line twice.)
// A plain function with by-value arguments: as a by-reference lambda the closure (and every captured local) ended up in scratch.
// Load number k (0 .. MT + 2 NF - 1) of a chunk.  `real` false (no chunk left to
// fetch): every lane reads the first 16 bytes of the X image into this wave's 1 KiB of scratch behind the stages — one L2 hit,
// no stage is touched (in the first step the "pending" half-chunk does not exist yet and every stage is live) — so that the
// number of loads per step, and with it every s_waitcnt vmcnt in the pipeline, is a constant and the step has no branches.
template <int MT, int NF, bool PK>
__device__ __forceinline__ void v3_issue_one(int k, bool real, const char* dummy_src, char* dummy_dst, char* sb, const char* xs,
                                             const char* wsrc /* PK: tile image + lane*16; else matrix + chunk column + piece */, int row0, int R, int C, int wave) {
    constexpr int XI = MT * 32 * 64 * 2;
    if (k < MT) { const char* src = xs + (wave + 4 * k) * 1024; glds16(real ? src : dummy_src, real ? sb + (wave + 4 * k) * 1024 : dummy_dst); return; }
    const int j = k - MT;
    char* const dst = sb + XI + (wave * 2 * NF + j) * 1024;
    if (PK) { const char* src = wsrc + (wave * 2 * NF + j) * 1024; glds16<2>(real ? src : dummy_src, real ? dst : dummy_dst); return; }
    int row = row0 + 16 * j; row = row < R ? row : R - 1;
    const char* src = reinterpret_cast<const char*>(reinterpret_cast<const _Float16*>(wsrc) + (size_t)row * C);
    glds16(real ? src : dummy_src, real ? dst : dummy_dst);
}

#ifdef THK_PREFILL_TRACE   // development build only (tools/dev/prefill_trace.py): per-wave cycle totals of the main loop's phases
__device__ unsigned long long g_pf_trace[4 * 256 * 4 * 12];   // [launch index mod 4][workgroup][wave][phase]
#define PF_T(i) { const unsigned long long now_ = __builtin_readcyclecounter(); tr_[i] += now_ - tlast_; tlast_ = now_; }
#else
#define PF_T(i)
#endif
template <int MT, int NF /* 32-row fragments per wave: 2 (256-row tile) or 1 (128-row tile) */, bool PK /* weights are tile images */, int NST = kNST>
__global__ __launch_bounds__(256, 1) void gemm_prefill_v3_kernel(const _Float16* __restrict__ w0, const _Float16* __restrict__ w1, const _Float16* __restrict__ w2,
                                                                 const char* __restrict__ ximg, const int nchunks, const int per, const int rb_per_mat, const int rb_total,
                                                                 const int R, const int C, float* __restrict__ part, const PrefillPlan plan) {
    // the ten leading scalars are everything the prologue needs before its first load: they arrive preloaded in SGPRs
    // (-amdgpu-kernarg-preload-count, see gemv_kernel in thk_kernels.hip); the rest of the plan is read when the tile is spilled
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    constexpr int XI = MT * 32 * 64 * 2;               // X image bytes per stage (hi rows, then lo rows)
    constexpr int TR = 128 * NF;                       // tile rows
    constexpr int WI = TR * 64;                        // W image: 4 waves x (32 NF) rows x 64 B
    constexpr int ST = XI + WI;
    constexpr int LPS = MT + 2 * NF;                   // loads per wave per stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31;
    // plain scalars (lambdas capturing the kernarg struct by reference push it to scratch)
    const int maxseg = plan.maxseg;
    const size_t slot_floats = plan.slot_floats;
    const int total = rb_total * nchunks;               // <= a few 10^4
    const int g0 = blockIdx.x * per;
    const int g1 = g0 + per < total ? g0 + per : total;
    if (g0 >= g1) return;

    // loader: lane -> (row lane>>2 of each 16-row group, position lane&3); the piece stored at that position
    const int ld_piece = swz_pos(lane >> 2, lane & 3);   // (row>>2)&3 is the same for rows r and r+16k
    // reader: fragment row li, piece 2*ks + (lane>>5)
    const int rd0 = li * 64 + swz_pos(li, lane >> 5) * 16, rd1 = li * 64 + swz_pos(li, 2 + (lane >> 5)) * 16;
    // issue cursor: (matrix, row-block in matrix, chunk) of the next flattened chunk to load
    const int rbk_first = g0 / nchunks;
    int i_mat = rbk_first / rb_per_mat, i_rbl = rbk_first % rb_per_mat, i_ch = g0 % nchunks, i_buf = 0;
    auto flush = [&](f16v (&acc)[NF][MT], int seg) __attribute__((always_inline)) {    // accumulators -> partial slot
        float* slot = part + ((size_t)blockIdx.x * maxseg + seg) * slot_floats;
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g)         // fragment order: one contiguous 1 KiB store per (f, t, g); frag_decode() below is the inverse
                    *reinterpret_cast<f4*>(slot + (size_t)(((((wave * NF + f) * MT + t) * 4 + g) * 64 + lane) * 4)) =
                        f4{acc[f][t][4 * g], acc[f][t][4 * g + 1], acc[f][t][4 * g + 2], acc[f][t][4 * g + 3]};   // (sc1 write-through stores: no gain, r02)
    };

    // ---- software pipeline across chunks --------------------------------------------------------------------------
    // While the matrix pipe works through chunk g out of REGISTERS (one fragment set), the same instruction stream, between
    // the MFMAs, (1) waits for chunk g+1's DMA and meets the other waves at the one barrier per chunk, (2) reads chunk g+1's
    // fragments into the other register set, (3) issues the DMA of chunk g+NST into the stage chunk g just vacated.
    // The r02 phase trace of the unpipelined loop (profiles/r02_prefill_phase_trace.txt): per chunk 1024 cycles of MFMA but
    // ~450 of DMA issue (all four waves hit the CU's one address unit together after the barrier), ~250 of exposed
    // fragment-read latency and ~300 of waitcnt + barrier, none of it overlapped with the matrix pipe (one wave per SIMD).
    char* const scratch = lds + NST * ST + wave * 1024;   // where dummy loads land
    int issued = g0;
#ifdef THK_PREFILL_TRACE
    unsigned long long tr_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast_ = __builtin_readcyclecounter();
    const unsigned long long tstart_ = tlast_, wstart_ = wall_clock64();
#endif
#pragma unroll
    for (int s = 0; s < NST; ++s) {                      // the first NST chunks (dummy loads where the share is shorter: constant counts)
        const bool more = issued < g1;
        const _Float16* nx_wm = i_mat == 0 ? w0 : (i_mat == 1 ? w1 : w2);
        const char* const nx_xs = ximg + (size_t)i_ch * XI + lane * 16;
        const char* const nx_w = PK ? reinterpret_cast<const char*>(nx_wm) + THK_WTILE(i_rbl, i_ch) * WI + lane * 16
                                    : reinterpret_cast<const char*>(nx_wm + (size_t)i_ch * kKC + ld_piece * 8);
#pragma unroll
        for (int k = 0; k < LPS; ++k) v3_issue_one<MT, NF, PK>(k, more, ximg, scratch, lds + s * ST, nx_xs, nx_w, i_rbl * TR + wave * 32 * NF + (lane >> 2), R, C, wave);
        if (more) {
            ++issued;
            if (++i_ch == nchunks) { i_ch = 0; if (++i_rbl == rb_per_mat) { i_rbl = 0; ++i_mat; } }
        }
    }
    i_buf = 0;                                           // NST loads went out: the next one refills stage 0
    // fragment registers: two sets (the chunk being multiplied, the chunk being fetched)
    h8 fa[2][2][NF], fbh[2][2][MT], fbl[2][2][MT];
    constexpr int NM = 4 * NF * MT;                      // MFMAs per chunk per wave
    constexpr int NR = 2 * (NF + 2 * MT);                // fragment reads per chunk per wave
    constexpr int SYNC_AT = NM >= 16 ? NM / 4 - 1 : 0;   // the wait + barrier sit behind this MFMA (late enough that chunk g+1 has normally landed)
    constexpr int SLOTS = NM - 1 - SYNC_AT;              // MFMAs behind the barrier: the loads are dealt out over them,
    constexpr int RSLOTS = SLOTS * 5 / 8 > 0 ? SLOTS * 5 / 8 : 1;   // the fragment reads over the first 5/8 (they must be back before the next step's first MFMA)
    // An LDS-DMA instruction costs its wave ~60 cycles of issue among bare MFMAs but 100-185 next to ds_read_b128 traffic
    // (MI355X_MICROARCH.md, timeline inputs), so the loads keep clear of the reads: the first LT of a chunk go out in the
    // slots behind the reads, the other LH in the NEXT step's MFMAs before its barrier.
    constexpr int TSLOTS = SLOTS - RSLOTS;
    constexpr int LT = SYNC_AT > 0 ? LPS / 2 : LPS, LH = LPS - LT;
    bool pend_more = false;                              // second half of the previous step's chunk (nothing in the first step: dummy loads)
    char* pend_sb = lds;
    const char *pend_xs = ximg, *pend_w = ximg;
    int pend_row0 = 0;
    f16v acc[NF][MT];
#define THK_PIN_ACC()   /* keep the accumulators in AccVGPRs: left alone the allocator parks tiles in VGPRs across the loop edge and moves them back before every MFMA */ \
    _Pragma("unroll") for (int f = 0; f < NF; ++f)                         \
    _Pragma("unroll") for (int t = 0; t < MT; ++t) asm volatile("" : "+a"(acc[f][t]));
#define THK_ZERO_ACC()                                                     \
    _Pragma("unroll") for (int f = 0; f < NF; ++f)                         \
    _Pragma("unroll") for (int t = 0; t < MT; ++t)                         \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[f][t][i] = 0.f;     \
    THK_PIN_ACC()
    int buf = 0;                                         // stage of the current chunk
    // fragment read number RR of a chunk (k-step, then the NF row fragments, then hi/lo of each token tile) into set S
#define THK_READ_ONE(SB, S, RR)                                                                                        \
    {                                                                                                                  \
        const int ks_ = (RR) / (NF + 2 * MT), q_ = (RR) % (NF + 2 * MT);                                               \
        const char* rp_ = (SB) + (ks_ == 0 ? rd0 : rd1);                                                               \
        if (q_ < NF) fa[S][ks_][q_ < NF ? q_ : 0] = *reinterpret_cast<const h8*>(rp_ + XI + (wave * NF + q_) * 2048);  \
        else if (((q_ - NF) & 1) == 0) fbh[S][ks_][q_ >= NF ? (q_ - NF) >> 1 : 0] = *reinterpret_cast<const h8*>(rp_ + ((q_ - NF) >> 1) * 2048); \
        else fbl[S][ks_][q_ >= NF ? (q_ - NF) >> 1 : 0] = *reinterpret_cast<const h8*>(rp_ + XI / 2 + ((q_ - NF) >> 1) * 2048); \
    }
    // One chunk: MFMAs from fragment set S; the next chunk's fragments go into set 1 - S.  (A macro expanded twice, not a
    // lambda taking the arrays: nested by-reference closures leave every captured array in scratch.)
#define THK_STEP(S)                                                                                                    \
    {                                                                                                                  \
        const bool more = issued < g1;                                 /* uniform */                                   \
        const _Float16* nx_wm = i_mat == 0 ? w0 : (i_mat == 1 ? w1 : w2);                                              \
        char* const nx_sb = lds + i_buf * ST;                                                                          \
        const char* const nx_xs = ximg + (size_t)i_ch * XI + lane * 16;                                                \
        const char* const nx_w = PK ? reinterpret_cast<const char*>(nx_wm) + THK_WTILE(i_rbl, i_ch) * WI + lane * 16 \
                                    : reinterpret_cast<const char*>(nx_wm + (size_t)i_ch * kKC + ld_piece * 8);        \
        const int nx_row0 = i_rbl * TR + wave * 32 * NF + (lane >> 2);                                                 \
        i_buf = i_buf + 1 == NST ? 0 : i_buf + 1;                                                                      \
        if (more) {                                                                                                    \
            ++issued;                                                                                                  \
            if (++i_ch == nchunks) { i_ch = 0; if (++i_rbl == rb_per_mat) { i_rbl = 0; ++i_mat; } }                    \
        }                                                                                                              \
        const int nbuf = buf + 1 == NST ? 0 : buf + 1;                                                                 \
        const char* const sbn = lds + nbuf * ST;                                                                       \
        THK_PIN_ACC()                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        _Pragma("unroll") for (int i = 0; i < NM; ++i) {                                                               \
            {   /* MFMA number i: k-step, hi|lo, then a sweep over all NF*MT accumulators, so that an accumulator is touched again only NF*MT \
                   MFMAs later (back-to-back MFMAs into one accumulator wait out the full pipeline depth: measured ~50 instead of 32 cycles each). \
                   Per accumulator the order stays ks0.hi ks0.lo ks1.hi ks1.lo: results do not depend on the sweep. */                    \
                const int ks = i / (2 * NF * MT), j = i % (2 * NF * MT), hl = j / (NF * MT), t = (j % (NF * MT)) / NF, f = j % NF; \
                if (hl == 0) acc[f][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][ks][f], fbh[S][ks][t], acc[f][t], 0, 0, 0); \
                else if (!kNoLo) acc[f][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][ks][f], fbl[S][ks][t], acc[f][t], 0, 0, 0); \
            }                                                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            if (i < SYNC_AT) {                                                                                         \
                _Pragma("unroll") for (int k = LT + i * LH / (SYNC_AT > 0 ? SYNC_AT : 1); k < LT + (i + 1) * LH / (SYNC_AT > 0 ? SYNC_AT : 1); ++k)            \
                    if (!kNoDma) v3_issue_one<MT, NF, PK>(k, pend_more, ximg, scratch, pend_sb, pend_xs, pend_w, pend_row0, R, C, wave);     \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            }                                                                                                          \
            if (i == SYNC_AT) {                                                                                        \
                PF_T(1)                                                                                                \
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* the fragment reads of the stage about to be refilled are back (hundreds of cycles old: free) - stated, not left to timing */ \
                wait_vmcnt<(NST - 2) * LPS>();                 /* this wave's loads of chunk g+1 have landed (NST-2 chunks, real or dummy, were issued behind them) ... */ \
                PF_T(2)                                                                                                \
                __builtin_amdgcn_s_barrier();                  /* ... and everybody's; every wave holds chunk g in registers: its stage is free */ \
                PF_T(3)                                                                                                \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            }                                                                                                          \
            if (i >= SYNC_AT && i < NM - 1) {                                                                          \
                const int sl = i - SYNC_AT;                    /* behind the last chunk these reads fetch stale bytes nobody uses */ \
                if (!kNoRead && sl < RSLOTS) { _Pragma("unroll") for (int rr = sl * NR / RSLOTS; rr < (sl + 1) * NR / RSLOTS; ++rr) THK_READ_ONE(sbn, 1 - S, rr) } \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
                if (sl >= RSLOTS) {                                                                                    \
                    _Pragma("unroll") for (int k = (sl - RSLOTS) * LT / TSLOTS; k < (sl - RSLOTS + 1) * LT / TSLOTS; ++k) \
                        if (!kNoDma) v3_issue_one<MT, NF, PK>(k, more, ximg, scratch, nx_sb, nx_xs, nx_w, nx_row0, R, C, wave);              \
                }                                                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            }                                                                                                          \
            if (i == SYNC_AT + RSLOTS - 1) { PF_T(8) }                                                                 \
        }                                                                                                              \
        buf = nbuf;                                                                                                    \
        pend_more = more; pend_sb = nx_sb; pend_xs = nx_xs; pend_w = nx_w; pend_row0 = nx_row0;                         \
        THK_PIN_ACC()                                                                                                  \
        PF_T(4)                                                                                                        \
        if (g + 1 == seg_end) {                            /* end of a row-block (or of the share): spill the tile */   \
            flush(acc, seg);                                                                                           \
            wait_vmcnt<0>();                                                                                           \
            PF_T(0)                               /* stores share the counter with the DMA queue: drain, then count afresh */ \
            THK_ZERO_ACC()                                                                                             \
            ++seg;                                                                                                     \
            seg_end = seg_end + nchunks < g1 ? seg_end + nchunks : g1;                                                 \
        }                                                                                                              \
    }
    int seg = 0, seg_end = (rbk_first + 1) * nchunks < g1 ? (rbk_first + 1) * nchunks : g1;
    THK_ZERO_ACC()
    PF_T(0)
    wait_vmcnt<(NST - 1) * LPS>();                       // chunk g0 is in
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) THK_READ_ONE(lds, 0, rr)
    if (kNoRead) {
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) THK_READ_ONE(lds, 1, rr)
    }
    for (int g = g0;;) {
        THK_STEP(0)
        if (++g >= g1) break;
        THK_STEP(1)
        if (++g >= g1) break;
    }
#ifdef THK_PREFILL_TRACE
    tr_[7] = __builtin_readcyclecounter() - tstart_;
    tr_[5] = wstart_; tr_[6] = wall_clock64();           // chip-global 100 MHz clock: when this wave started / ended
    if (lane == 0) for (int i = 0; i < 12; ++i) g_pf_trace[((((plan.packed >> 8) & 3) * 256 + blockIdx.x) * 4 + wave) * 12 + i] = tr_[i];
#endif
#undef THK_STEP
#undef THK_READ_ONE
#undef THK_ZERO_ACC
#undef THK_PIN_ACC
}

// ---------------------------------------------------------------- 129 .. 256 tokens per weight pass (round 5)
// The same kernel for a slab of EIGHT token tiles: every weight chunk that reaches LDS is multiplied against 256 tokens instead of 128, so a long prompt
// pays half the weight passes, pipeline fills, tile spills and reducer launches per token (prompt time = 3.7 ms + 0.023 ms x tokens per 128-token slab,
// section 4.3: the fixed part is what a second token half rides on).  All 256 AccVGPRs hold the tile (NF = 2), so the fragment sets stay those of FOUR token
// tiles and a chunk becomes two half-steps of 32 MFMAs:
//   step A  tiles 0-3 from set 0;  between its MFMAs: the pending half of the previous refill's DMA, and the chunk's tiles 4-7 -> set 1 (same stage: no wait)
//   step B  tiles 4-7 from set 1;  wait + the one barrier per chunk, the NEXT chunk's tiles 0-3 -> set 0, the DMA of chunk g + NST into the vacated stage
// Stage = 32 KB of X image + 16 KB of W: three stages (148 KB of LDS).  Loads per chunk and every vmcnt count are constants, as in the 128-token kernel.
constexpr int kNSTH = 3;
template <int NF, bool PK, int NST = kNSTH>
__global__ __launch_bounds__(256, 1) void gemm_prefill_v3h_kernel(const _Float16* __restrict__ w0, const _Float16* __restrict__ w1, const _Float16* __restrict__ w2,
                                                                  const char* __restrict__ ximg, const int nchunks, const int per, const int rb_per_mat, const int rb_total,
                                                                  const int R, const int C, float* __restrict__ part, const PrefillPlan plan) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    constexpr int MT = 8, MTH = 4;
    constexpr int XI = MT * 32 * 64 * 2;               // X image bytes per stage (hi rows of the 256 tokens, then lo rows)
    constexpr int TR = 128 * NF;
    constexpr int WI = TR * 64;
    constexpr int ST = XI + WI;
    constexpr int LPS = MT + 2 * NF;                   // loads per wave per stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31;
    const int maxseg = plan.maxseg;
    const size_t slot_floats = plan.slot_floats;
    const int total = rb_total * nchunks;
    const int g0 = blockIdx.x * per;
    const int g1 = g0 + per < total ? g0 + per : total;
    if (g0 >= g1) return;
    const int ld_piece = swz_pos(lane >> 2, lane & 3);
    const int rd0 = li * 64 + swz_pos(li, lane >> 5) * 16, rd1 = li * 64 + swz_pos(li, 2 + (lane >> 5)) * 16;
    const int rbk_first = g0 / nchunks;
    int i_mat = rbk_first / rb_per_mat, i_rbl = rbk_first % rb_per_mat, i_ch = g0 % nchunks, i_buf = 0;
    auto flush = [&](f16v (&acc)[NF][MT], int seg) __attribute__((always_inline)) {    // accumulators -> partial slot, fragment order (frag_decode with MT = 8)
        float* slot = part + ((size_t)blockIdx.x * maxseg + seg) * slot_floats;
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f4*>(slot + (size_t)(((((wave * NF + f) * MT + t) * 4 + g) * 64 + lane) * 4)) =
                        f4{acc[f][t][4 * g], acc[f][t][4 * g + 1], acc[f][t][4 * g + 2], acc[f][t][4 * g + 3]};
    };
    char* const scratch = lds + NST * ST + wave * 1024;   // where dummy loads land
    int issued = g0;
#pragma unroll
    for (int s = 0; s < NST; ++s) {                      // the first NST chunks (dummy loads where the share is shorter: constant counts)
        const bool more = issued < g1;
        const _Float16* nx_wm = i_mat == 0 ? w0 : (i_mat == 1 ? w1 : w2);
        const char* const nx_xs = ximg + (size_t)i_ch * XI + lane * 16;
        const char* const nx_w = PK ? reinterpret_cast<const char*>(nx_wm) + THK_WTILE(i_rbl, i_ch) * WI + lane * 16
                                    : reinterpret_cast<const char*>(nx_wm + (size_t)i_ch * kKC + ld_piece * 8);
#pragma unroll
        for (int k = 0; k < LPS; ++k) v3_issue_one<MT, NF, PK>(k, more, ximg, scratch, lds + s * ST, nx_xs, nx_w, i_rbl * TR + wave * 32 * NF + (lane >> 2), R, C, wave);
        if (more) {
            ++issued;
            if (++i_ch == nchunks) { i_ch = 0; if (++i_rbl == rb_per_mat) { i_rbl = 0; ++i_mat; } }
        }
    }
    i_buf = 0;
    h8 fa[2][2][NF], fbh[2][2][MTH], fbl[2][2][MTH];     // two sets: one per token HALF (set 0 = tiles 0-3, set 1 = tiles 4-7)
    constexpr int NM = 4 * NF * MTH;                     // MFMAs per half-step per wave
    constexpr int NRH = 2 * (NF + 2 * MTH);              // fragment reads per half-step per wave
    constexpr int SYNC_AT = NM / 4 - 1;                  // step B: the wait + barrier sit behind this MFMA
    constexpr int SLOTS = NM - 1 - SYNC_AT;
    constexpr int RSLOTS = SLOTS * 5 / 8 > 0 ? SLOTS * 5 / 8 : 1;
    constexpr int TSLOTS = SLOTS - RSLOTS;
    constexpr int LT = LPS / 2, LH = LPS - LT;           // a refill: LT loads in step B's tail, LH in the next step A's head
    constexpr int AL = NM / 4;                           // step A: slots that carry the pending LH loads ...
    constexpr int ARS = NM - AL - NM / 8;                // ... and the slots behind them that carry the reads of tiles 4-7 (back before step B's first MFMA)
    bool pend_more = false;
    char* pend_sb = lds;
    const char *pend_xs = ximg, *pend_w = ximg;
    int pend_row0 = 0;
    f16v acc[NF][MT];
#define THK_PIN_ACC()                                                      \
    _Pragma("unroll") for (int f = 0; f < NF; ++f)                         \
    _Pragma("unroll") for (int t = 0; t < MT; ++t) asm volatile("" : "+a"(acc[f][t]));
#define THK_ZERO_ACC()                                                     \
    _Pragma("unroll") for (int f = 0; f < NF; ++f)                         \
    _Pragma("unroll") for (int t = 0; t < MT; ++t)                         \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[f][t][i] = 0.f;     \
    THK_PIN_ACC()
    int buf = 0;
    // fragment read number RR of a half-step (k-step, then the NF row fragments, then hi/lo of the half's four token tiles) into set S; H = token half
#define THK_READ_ONE(SB, S, RR, H)                                                                                     \
    {                                                                                                                  \
        const int ks_ = (RR) / (NF + 2 * MTH), q_ = (RR) % (NF + 2 * MTH);                                             \
        const char* rp_ = (SB) + (ks_ == 0 ? rd0 : rd1);                                                               \
        if (q_ < NF) fa[S][ks_][q_ < NF ? q_ : 0] = *reinterpret_cast<const h8*>(rp_ + XI + (wave * NF + q_) * 2048);  \
        else if (((q_ - NF) & 1) == 0) fbh[S][ks_][q_ >= NF ? (q_ - NF) >> 1 : 0] = *reinterpret_cast<const h8*>(rp_ + (4 * (H) + ((q_ - NF) >> 1)) * 2048); \
        else fbl[S][ks_][q_ >= NF ? (q_ - NF) >> 1 : 0] = *reinterpret_cast<const h8*>(rp_ + XI / 2 + (4 * (H) + ((q_ - NF) >> 1)) * 2048); \
    }
#define THK_MFMA_ONE(S, I, TOFF)                                                                                       \
    {                                                                                                                  \
        const int ks = (I) / (2 * NF * MTH), j = (I) % (2 * NF * MTH), hl = j / (NF * MTH), t = (j % (NF * MTH)) / NF, f = j % NF; \
        if (hl == 0) acc[f][(TOFF) + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][ks][f], fbh[S][ks][t], acc[f][(TOFF) + t], 0, 0, 0); \
        else if (!kNoLo) acc[f][(TOFF) + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][ks][f], fbl[S][ks][t], acc[f][(TOFF) + t], 0, 0, 0); \
    }
    // step A: tiles 0-3 from set 0; pending DMA half; this chunk's tiles 4-7 -> set 1
#define THK_STEP_A()                                                                                                   \
    {                                                                                                                  \
        const char* const sbc = lds + buf * ST;                                                                        \
        THK_PIN_ACC()                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        _Pragma("unroll") for (int i = 0; i < NM; ++i) {                                                               \
            THK_MFMA_ONE(0, i, 0)                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            if (i < AL) {                                                                                              \
                _Pragma("unroll") for (int k = LT + i * LH / AL; k < LT + (i + 1) * LH / AL; ++k)                      \
                    v3_issue_one<MT, NF, PK>(k, pend_more, ximg, scratch, pend_sb, pend_xs, pend_w, pend_row0, R, C, wave); \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            } else if (i < AL + ARS) {                                                                                 \
                _Pragma("unroll") for (int rr = (i - AL) * NRH / ARS; rr < (i - AL + 1) * NRH / ARS; ++rr) THK_READ_ONE(sbc, 1, rr, 1) \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            }                                                                                                          \
        }                                                                                                              \
        THK_PIN_ACC()                                                                                                  \
    }
    // step B: tiles 4-7 from set 1; wait + barrier; the next chunk's tiles 0-3 -> set 0; refill the vacated stage
#define THK_STEP_B()                                                                                                   \
    {                                                                                                                  \
        const bool more = issued < g1;                                 /* uniform */                                   \
        const _Float16* nx_wm = i_mat == 0 ? w0 : (i_mat == 1 ? w1 : w2);                                              \
        char* const nx_sb = lds + i_buf * ST;                                                                          \
        const char* const nx_xs = ximg + (size_t)i_ch * XI + lane * 16;                                                \
        const char* const nx_w = PK ? reinterpret_cast<const char*>(nx_wm) + THK_WTILE(i_rbl, i_ch) * WI + lane * 16 \
                                    : reinterpret_cast<const char*>(nx_wm + (size_t)i_ch * kKC + ld_piece * 8);        \
        const int nx_row0 = i_rbl * TR + wave * 32 * NF + (lane >> 2);                                                 \
        i_buf = i_buf + 1 == NST ? 0 : i_buf + 1;                                                                      \
        if (more) {                                                                                                    \
            ++issued;                                                                                                  \
            if (++i_ch == nchunks) { i_ch = 0; if (++i_rbl == rb_per_mat) { i_rbl = 0; ++i_mat; } }                    \
        }                                                                                                              \
        const int nbuf = buf + 1 == NST ? 0 : buf + 1;                                                                 \
        const char* const sbn = lds + nbuf * ST;                                                                       \
        THK_PIN_ACC()                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        _Pragma("unroll") for (int i = 0; i < NM; ++i) {                                                               \
            THK_MFMA_ONE(1, i, MTH)                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            if (i == SYNC_AT) {                                                                                        \
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* step A's reads of tiles 4-7 from the stage about to be refilled are back (hundreds of cycles old: free) */ \
                wait_vmcnt<(NST - 2) * LPS>();                 /* this wave's loads of chunk g+1 have landed ... */    \
                __builtin_amdgcn_s_barrier();                  /* ... and everybody's; every wave holds chunk g's second half in registers: its stage is free */ \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            }                                                                                                          \
            if (i >= SYNC_AT && i < NM - 1) {                                                                          \
                const int sl = i - SYNC_AT;                    /* behind the last chunk these reads fetch stale bytes nobody uses */ \
                if (sl < RSLOTS) { _Pragma("unroll") for (int rr = sl * NRH / RSLOTS; rr < (sl + 1) * NRH / RSLOTS; ++rr) THK_READ_ONE(sbn, 0, rr, 0) } \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
                if (sl >= RSLOTS) {                                                                                    \
                    _Pragma("unroll") for (int k = (sl - RSLOTS) * LT / TSLOTS; k < (sl - RSLOTS + 1) * LT / TSLOTS; ++k) \
                        v3_issue_one<MT, NF, PK>(k, more, ximg, scratch, nx_sb, nx_xs, nx_w, nx_row0, R, C, wave);     \
                }                                                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            }                                                                                                          \
        }                                                                                                              \
        buf = nbuf;                                                                                                    \
        pend_more = more; pend_sb = nx_sb; pend_xs = nx_xs; pend_w = nx_w; pend_row0 = nx_row0;                         \
        THK_PIN_ACC()                                                                                                  \
        if (g + 1 == seg_end) {                            /* end of a row-block (or of the share): spill the tile */   \
            flush(acc, seg);                                                                                           \
            wait_vmcnt<0>();                               /* stores share the counter with the DMA queue: drain, then count afresh */ \
            THK_ZERO_ACC()                                                                                             \
            ++seg;                                                                                                     \
            seg_end = seg_end + nchunks < g1 ? seg_end + nchunks : g1;                                                 \
        }                                                                                                              \
    }
    int seg = 0, seg_end = (rbk_first + 1) * nchunks < g1 ? (rbk_first + 1) * nchunks : g1;
    THK_ZERO_ACC()
    wait_vmcnt<(NST - 1) * LPS>();                       // chunk g0 is in
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int rr = 0; rr < NRH; ++rr) THK_READ_ONE(lds, 0, rr, 0)
    for (int g = g0;;) {
        THK_STEP_A()
        THK_STEP_B()
        if (++g >= g1) break;
    }
#undef THK_STEP_A
#undef THK_STEP_B
#undef THK_MFMA_ONE
#undef THK_READ_ONE
#undef THK_ZERO_ACC
#undef THK_PIN_ACC
}

// ---------------------------------------------------------------- 256-token slabs on a 2 x 2 WAVE GRID (round 6)
// gemm_prefill_v3h_kernel gives every wave its own ROWS and all 256 tokens: a wave reads every token fragment of the chunk, so the X image leaves LDS four
// times per chunk - per k-step and wave NF + 16 fragment reads (18 for a 256-row tile) for 16 NF MFMAs.  Here wave (wr, wt) owns HALF the tile's rows
// (RF = 2 NF fragments) and HALF the slab's tokens (four tiles): a W fragment is read by two waves, an X fragment by two instead of four - RF + 8 reads per
// k-step (12 for a 256-row tile, 10 for a 128-row one) for the same 8 RF MFMAs, same 64 NF AccVGPRs, same stages, same DMA schedule, same partial-slot format.
// A chunk is still two half-steps, now split by K-STEP instead of by token half (one fragment set = one k-step of the wave's 2 x 2 block: 48 VGPRs):
//   step A  k-step 0 from set 0;  between its MFMAs: the pending half of the previous refill's DMA, and the chunk's k-step 1 -> set 1 (same stage: no wait)
//   step B  k-step 1 from set 1;  wait + the one barrier per chunk, the NEXT chunk's k-step 0 -> set 0, the DMA of chunk g + NST into the vacated stage
// Per accumulator the order stays ks0.hi ks0.lo ks1.hi ks1.lo: bit-identical to gemm_prefill_v3h_kernel.
template <int NF, bool PK, int NST = kNSTH>
__global__ __launch_bounds__(256, 1) void gemm_prefill_v3g_kernel(const _Float16* __restrict__ w0, const _Float16* __restrict__ w1, const _Float16* __restrict__ w2,
                                                                  const char* __restrict__ ximg, const int nchunks, const int per, const int rb_per_mat, const int rb_total,
                                                                  const int R, const int C, float* __restrict__ part, const PrefillPlan plan) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    constexpr int MT = 8, TT = 4;                      // token tiles of the slab / of a wave
    constexpr int RF = 2 * NF;                         // row fragments of a wave (half the tile)
    constexpr int XI = MT * 32 * 64 * 2;
    constexpr int TR = 128 * NF;
    constexpr int WI = TR * 64;
    constexpr int ST = XI + WI;
    constexpr int LPS = MT + 2 * NF;                   // loads per wave per stage (the loader's split of a stage is the v3h one: who fetches is independent of who reads)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wt = wave & 1;           // row half, token half
    const int li = lane & 31;
    const int maxseg = plan.maxseg;
    const size_t slot_floats = plan.slot_floats;
    const int total = rb_total * nchunks;
    const int g0 = blockIdx.x * per;
    const int g1 = g0 + per < total ? g0 + per : total;
    if (g0 >= g1) return;
    const int ld_piece = swz_pos(lane >> 2, lane & 3);
    const int rd0 = li * 64 + swz_pos(li, lane >> 5) * 16, rd1 = li * 64 + swz_pos(li, 2 + (lane >> 5)) * 16;
    const int rbk_first = g0 / nchunks;
    int i_mat = rbk_first / rb_per_mat, i_rbl = rbk_first % rb_per_mat, i_ch = g0 % nchunks, i_buf = 0;
    auto flush = [&](f16v (&acc)[RF][TT], int seg) __attribute__((always_inline)) {    // accumulators -> partial slot, the fragment order of the other kernels (frag_decode with MT = 8)
        float* slot = part + ((size_t)blockIdx.x * maxseg + seg) * slot_floats;
#pragma unroll
        for (int f = 0; f < RF; ++f)
#pragma unroll
            for (int t = 0; t < TT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f4*>(slot + (size_t)(((((wr * RF + f) * MT + (wt * TT + t)) * 4 + g) * 64 + lane) * 4)) =
                        f4{acc[f][t][4 * g], acc[f][t][4 * g + 1], acc[f][t][4 * g + 2], acc[f][t][4 * g + 3]};
    };
    char* const scratch = lds + NST * ST + wave * 1024;   // where dummy loads land
    int issued = g0;
#pragma unroll
    for (int s = 0; s < NST; ++s) {                      // the first NST chunks (dummy loads where the share is shorter: constant counts)
        const bool more = issued < g1;
        const _Float16* nx_wm = i_mat == 0 ? w0 : (i_mat == 1 ? w1 : w2);
        const char* const nx_xs = ximg + (size_t)i_ch * XI + lane * 16;
        const char* const nx_w = PK ? reinterpret_cast<const char*>(nx_wm) + THK_WTILE(i_rbl, i_ch) * WI + lane * 16
                                    : reinterpret_cast<const char*>(nx_wm + (size_t)i_ch * kKC + ld_piece * 8);
#pragma unroll
        for (int k = 0; k < LPS; ++k) v3_issue_one<MT, NF, PK>(k, more, ximg, scratch, lds + s * ST, nx_xs, nx_w, i_rbl * TR + wave * 32 * NF + (lane >> 2), R, C, wave);
        if (more) {
            ++issued;
            if (++i_ch == nchunks) { i_ch = 0; if (++i_rbl == rb_per_mat) { i_rbl = 0; ++i_mat; } }
        }
    }
    i_buf = 0;
    h8 fa[2][RF], fbh[2][TT], fbl[2][TT];               // two sets: one per K-STEP of a chunk
    constexpr int NM = 2 * RF * TT;                      // MFMAs per half-step per wave (hi and lo of one k-step)
    constexpr int NRH = RF + 2 * TT;                     // fragment reads per half-step per wave
    constexpr int SYNC_AT = NM / 4 - 1;                  // step B: the wait + barrier sit behind this MFMA
    constexpr int SLOTS = NM - 1 - SYNC_AT;
    constexpr int RSLOTS = SLOTS * 5 / 8 > 0 ? SLOTS * 5 / 8 : 1;
    constexpr int TSLOTS = SLOTS - RSLOTS;
    constexpr int LT = LPS / 2, LH = LPS - LT;           // a refill: LT loads in step B's tail, LH in the next step A's head
    constexpr int AL = NM / 4;                           // step A: slots that carry the pending LH loads ...
    constexpr int ARS = NM - AL - NM / 8;                // ... and the slots behind them that carry the reads of k-step 1 (back before step B's first MFMA)
    bool pend_more = false;
    char* pend_sb = lds;
    const char *pend_xs = ximg, *pend_w = ximg;
    int pend_row0 = 0;
    f16v acc[RF][TT];
#define THK_PIN_ACC()                                                      \
    _Pragma("unroll") for (int f = 0; f < RF; ++f)                         \
    _Pragma("unroll") for (int t = 0; t < TT; ++t) asm volatile("" : "+a"(acc[f][t]));
#define THK_ZERO_ACC()                                                     \
    _Pragma("unroll") for (int f = 0; f < RF; ++f)                         \
    _Pragma("unroll") for (int t = 0; t < TT; ++t)                         \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[f][t][i] = 0.f;     \
    THK_PIN_ACC()
    int buf = 0;
    // fragment read number RR of k-step KS (the wave's RF row fragments, then hi/lo of its four token tiles) into set S
#define THK_READ_ONE(SB, S, RR, KS)                                                                                    \
    {                                                                                                                  \
        const char* rp_ = (SB) + ((KS) == 0 ? rd0 : rd1);                                                              \
        if ((RR) < RF) fa[S][(RR) < RF ? (RR) : 0] = *reinterpret_cast<const h8*>(rp_ + XI + (wr * RF + (RR)) * 2048); \
        else if ((((RR) - RF) & 1) == 0) fbh[S][(RR) >= RF ? ((RR) - RF) >> 1 : 0] = *reinterpret_cast<const h8*>(rp_ + (wt * TT + (((RR) - RF) >> 1)) * 2048); \
        else fbl[S][(RR) >= RF ? ((RR) - RF) >> 1 : 0] = *reinterpret_cast<const h8*>(rp_ + XI / 2 + (wt * TT + (((RR) - RF) >> 1)) * 2048); \
    }
    // MFMA number I of a half-step: hi | lo, then a sweep over all RF x TT accumulators (an accumulator is touched again only RF * TT MFMAs later)
#define THK_MFMA_ONE(S, I)                                                                                             \
    {                                                                                                                  \
        const int hl = (I) / (RF * TT), t = ((I) % (RF * TT)) / RF, f = (I) % RF;                                      \
        if (hl == 0) acc[f][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][f], fbh[S][t], acc[f][t], 0, 0, 0);      \
        else if (!kNoLo) acc[f][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][f], fbl[S][t], acc[f][t], 0, 0, 0);  \
    }
    // step A: k-step 0 from set 0; pending DMA half; this chunk's k-step 1 -> set 1
#define THK_STEP_A()                                                                                                   \
    {                                                                                                                  \
        const char* const sbc = lds + buf * ST;                                                                        \
        THK_PIN_ACC()                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        _Pragma("unroll") for (int i = 0; i < NM; ++i) {                                                               \
            THK_MFMA_ONE(0, i)                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            if (i < AL) {                                                                                              \
                _Pragma("unroll") for (int k = LT + i * LH / AL; k < LT + (i + 1) * LH / AL; ++k)                      \
                    v3_issue_one<MT, NF, PK>(k, pend_more, ximg, scratch, pend_sb, pend_xs, pend_w, pend_row0, R, C, wave); \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            } else if (i < AL + ARS) {                                                                                 \
                _Pragma("unroll") for (int rr = (i - AL) * NRH / ARS; rr < (i - AL + 1) * NRH / ARS; ++rr) THK_READ_ONE(sbc, 1, rr, 1) \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            }                                                                                                          \
        }                                                                                                              \
        THK_PIN_ACC()                                                                                                  \
    }
    // step B: k-step 1 from set 1; wait + barrier; the next chunk's k-step 0 -> set 0; refill the vacated stage
#define THK_STEP_B()                                                                                                   \
    {                                                                                                                  \
        const bool more = issued < g1;                                 /* uniform */                                   \
        const _Float16* nx_wm = i_mat == 0 ? w0 : (i_mat == 1 ? w1 : w2);                                              \
        char* const nx_sb = lds + i_buf * ST;                                                                          \
        const char* const nx_xs = ximg + (size_t)i_ch * XI + lane * 16;                                                \
        const char* const nx_w = PK ? reinterpret_cast<const char*>(nx_wm) + THK_WTILE(i_rbl, i_ch) * WI + lane * 16 \
                                    : reinterpret_cast<const char*>(nx_wm + (size_t)i_ch * kKC + ld_piece * 8);        \
        const int nx_row0 = i_rbl * TR + wave * 32 * NF + (lane >> 2);                                                 \
        i_buf = i_buf + 1 == NST ? 0 : i_buf + 1;                                                                      \
        if (more) {                                                                                                    \
            ++issued;                                                                                                  \
            if (++i_ch == nchunks) { i_ch = 0; if (++i_rbl == rb_per_mat) { i_rbl = 0; ++i_mat; } }                    \
        }                                                                                                              \
        const int nbuf = buf + 1 == NST ? 0 : buf + 1;                                                                 \
        const char* const sbn = lds + nbuf * ST;                                                                       \
        THK_PIN_ACC()                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        _Pragma("unroll") for (int i = 0; i < NM; ++i) {                                                               \
            THK_MFMA_ONE(1, i)                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            if (i == SYNC_AT) {                                                                                        \
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* this wave's k-step-1 reads of the stage about to be refilled are back (they are hundreds of cycles old: free) */ \
                wait_vmcnt<(NST - 2) * LPS>();                 /* this wave's loads of chunk g+1 have landed ... */    \
                __builtin_amdgcn_s_barrier();                  /* ... and everybody's; every wave holds chunk g's second k-step in registers: its stage is free */ \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            }                                                                                                          \
            if (i >= SYNC_AT && i < NM - 1) {                                                                          \
                const int sl = i - SYNC_AT;                    /* behind the last chunk these reads fetch stale bytes nobody uses */ \
                if (sl < RSLOTS) { _Pragma("unroll") for (int rr = sl * NRH / RSLOTS; rr < (sl + 1) * NRH / RSLOTS; ++rr) THK_READ_ONE(sbn, 0, rr, 0) } \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
                if (sl >= RSLOTS) {                                                                                    \
                    _Pragma("unroll") for (int k = (sl - RSLOTS) * LT / TSLOTS; k < (sl - RSLOTS + 1) * LT / TSLOTS; ++k) \
                        v3_issue_one<MT, NF, PK>(k, more, ximg, scratch, nx_sb, nx_xs, nx_w, nx_row0, R, C, wave);     \
                }                                                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            }                                                                                                          \
        }                                                                                                              \
        buf = nbuf;                                                                                                    \
        pend_more = more; pend_sb = nx_sb; pend_xs = nx_xs; pend_w = nx_w; pend_row0 = nx_row0;                         \
        THK_PIN_ACC()                                                                                                  \
        if (g + 1 == seg_end) {                            /* end of a row-block (or of the share): spill the tile */   \
            flush(acc, seg);                                                                                           \
            wait_vmcnt<0>();                               /* stores share the counter with the DMA queue: drain, then count afresh */ \
            THK_ZERO_ACC()                                                                                             \
            ++seg;                                                                                                     \
            seg_end = seg_end + nchunks < g1 ? seg_end + nchunks : g1;                                                 \
        }                                                                                                              \
    }
    int seg = 0, seg_end = (rbk_first + 1) * nchunks < g1 ? (rbk_first + 1) * nchunks : g1;
    THK_ZERO_ACC()
    wait_vmcnt<(NST - 1) * LPS>();                       // chunk g0 is in
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int rr = 0; rr < NRH; ++rr) THK_READ_ONE(lds, 0, rr, 0)
    for (int g = g0;;) {
        THK_STEP_A()
        THK_STEP_B()
        if (++g >= g1) break;
    }
#undef THK_STEP_A
#undef THK_STEP_B
#undef THK_MFMA_ONE
#undef THK_READ_ONE
#undef THK_ZERO_ACC
#undef THK_PIN_ACC
}

static_assert(KernargLead<decltype(&gemm_prefill_v3_kernel<4, 2, true, kNST>)>::bytes() == kKernargPreloadBytes + 8,
              "gemm_prefill_v3_kernel: ten scalars (14 dwords, preloaded) and the partial-tile pointer ahead of the plan");

// ---- weight tile images ------------------------------------------------------------------------
// A row-major f16 matrix [R][C] rewritten as the sequence of LDS images the GEMM loads: for row-block rb (tile_rows rows)
// and K-chunk ch (32 columns) one contiguous block of tile_rows x 64 bytes, row r_in_tile at r_in_tile*64, its four
// 16-byte pieces at position  piece ^ ((r_in_tile >> 2) & 3)  (the swizzle of swz_pos).  Rows past R are zeros.
// One thread per 16-byte piece; reads are row-contiguous, writes are 64-byte segments.
__global__ __launch_bounds__(256) void pack_w_kernel(const _Float16* __restrict__ W, int R, int C, int TR, int Rpad, char* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int ppr = C >> 3;                                   // pieces per row
    if (idx >= (size_t)Rpad * ppr) return;
    const int row = (int)(idx / ppr), pc = (int)(idx % ppr);
    const int rb = row / TR, rit = row % TR, ch = pc >> 2, piece = pc & 3;
    h8 v = h8{0, 0, 0, 0, 0, 0, 0, 0};
    if (row < R) v = *reinterpret_cast<const h8*>(W + (size_t)row * C + (size_t)pc * 8);
    *reinterpret_cast<h8*>(out + (((size_t)rb * (C / kKC) + ch) * TR + rit) * 64 + (size_t)swz_pos(rit, piece) * 16) = v;
}
size_t prefill_pack_bytes(int R, int C, int tile_rows) {
    const int TR = tile_rows == 128 ? 128 : kV3Rows;
    return (size_t)((R + TR - 1) / TR * TR) * C * 2;
}
hipError_t launch_prefill_pack(const uint16_t* W, int R, int C, int tile_rows, void* out, hipStream_t st) {
    if (R < 1 || C % kKC != 0) return hipErrorInvalidValue;
    const int TR = tile_rows == 128 ? 128 : kV3Rows, Rpad = (R + TR - 1) / TR * TR;
    const size_t n = (size_t)Rpad * (C / 8);
    hipLaunchKernelGGL(pack_w_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const _Float16*>(W), R, C, TR, Rpad, (char*)out);
    return hipGetLastError();
}

// ---- X-image writers -------------------------------------------------------------------------
// byte offset of the 16-byte piece (token, 8 columns starting at col) inside the image
__device__ __forceinline__ size_t ximg_off(int MT, int arr, int tok, int col) {
    return (size_t)(col >> 5) * ximg_stage_bytes(MT) + (size_t)arr * MT * 32 * 64 + (size_t)tok * 64 + (size_t)(((col >> 3) & 3) ^ ((tok >> 2) & 3)) * 16;
}
__device__ __forceinline__ void ximg_store8(char* img, int MT, int tok, int col, const float (&v)[8]) {
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)v[e]; lo[e] = (_Float16)(v[e] - (float)hi[e]); }
    *reinterpret_cast<h8*>(img + ximg_off(MT, 0, tok, col)) = hi;
    *reinterpret_cast<h8*>(img + ximg_off(MT, 1, tok, col)) = lo;
}
// Deferred RMSNorm (round 4).  x -> (x * inv) * g -> W is computed as inv * (W (x * g)): the per-token scalar inv = 1/sqrt(mean
// x^2 + eps) moves to the OUTPUT side of the GEMM, so the kernel that produces x (a reducer with the residual add) can write the
// hi/lo image of x * g itself - it does not have to know the whole row's sum of squares, which lives in other workgroups - and
// the norm -> image launch between two GEMMs disappears.  The sum of squares travels as 64-bit FIXED POINT (2^-32 units): every
// producer workgroup adds its share with an integer atomic, so the total does not depend on the order of arrival (bit-repeatable
// results, no float atomics), and the consuming reducer reads one word per token.
// The image still has to hold O(1) values: the lo half of the hi/lo split is an f16 too, and for |v| below ~0.1 it falls into
// f16's subnormals (a residual stream of magnitude 0.02 - the embedding scale - lost a factor 4 of accuracy that way).  So the
// producer multiplies by a POWER OF TWO near the token's 1/rms - exact, no rounding - taken from a sum of squares it CAN know:
// the token's previous norm input (the residual stream moves slowly), or the row's own for the first layer; the consumer divides
// it out again (it reads the same word and derives the same power of two).
constexpr float kSsqScale = 4294967296.f;    // 2^32: a 64-bit word holds sums up to 2^32, and a workgroup's share of a row with rms 1e-4 still has four digits
                                             // (2^24, rounds 3-4, left percent-level errors in 1/rms for residual streams of rms < 1e-3; advisor, round 4)
// RANGE: a token's sum of squares must stay below 2^32 (rms < 1024 at E = 4096, < 2896 at E = 512; LLaMA's residual stream with its outlier channels is
// three orders of magnitude below) - the integer atomic would wrap silently beyond it.  One share is clamped to 2^62 so that a single huge row saturates
// instead of wrapping; rows that large belong on prefill_deferred_norm = 0 (tests: test_prefill_deferred_norm_range).
__device__ __forceinline__ unsigned long long ssq_fixed(float ss) { return (unsigned long long)__float2ull_rn(fminf(ss * kSsqScale, 4.611686e18f)); }
__device__ __forceinline__ void ssq_add(unsigned long long* ssq, int tok, float ss) { atomicAdd(ssq + tok, ssq_fixed(ss)); }
__device__ __forceinline__ float ssq_inv_of(unsigned long long v, int C) {
    const float ss = (float)((double)v * (1.0 / 4294967296.0));
    return 1.0f / sqrtf(ss / (float)C + 1e-6f);
}
__device__ __forceinline__ float ssq_inv(const unsigned long long* ssq, int tok, int C) { return ssq_inv_of(ssq[tok], C); }
__device__ __forceinline__ float pow2_below(float v) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0x7F800000u); }   // v > 0, normal
__device__ __forceinline__ float ssq_pow2(const unsigned long long* ssq, int tok, int C) { return pow2_below(ssq_inv(ssq, tok, C)); }
// one workgroup per token row (pad rows are written as zeros).  MODE 1: v = (x * rsqrt(mean(x^2)+eps)) * gain  (K4+K5);
// MODE 2: v = x * 2^k * gain with 2^k <= 1/rms < 2^(k+1), the row's sum of squares goes to ssq[tok] (deferred norm; ssq[tok] must be zero)
template <int MODE>
__global__ __launch_bounds__(256) void ximg_from_rows_kernel(const float* __restrict__ X, const float* __restrict__ gain, int M, int MT, int C, char* __restrict__ img,
                                                             unsigned long long* __restrict__ ssq) {
    constexpr bool NORM = MODE == 1;
    __shared__ float red[4];
    const int tok = blockIdx.x;
    const float* row = X + (size_t)tok * C;
    float inv = 1.f;
    if (MODE != 0) {
        float ss = 0.f;
        if (tok < M) for (int i = threadIdx.x; i < C; i += 256) ss += row[i] * row[i];
        ss = wave_sum_prefill(ss);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
        __syncthreads();
        ss = (red[0] + red[1]) + (red[2] + red[3]);
        if (NORM) inv = 1.0f / sqrtf(ss / (float)C + 1e-6f);
        else {
            inv = pow2_below(ssq_inv_of(ssq_fixed(ss), C));      // what the consumer derives from the word written below
            if (threadIdx.x == 0 && tok < M) ssq_add(ssq, tok, ss);
        }
    }
    for (int col = threadIdx.x * 8; col < C; col += 256 * 8) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if (tok < M) {
            const f4 x0 = *reinterpret_cast<const f4*>(row + col), x1 = *reinterpret_cast<const f4*>(row + col + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = x0[e]; v[4 + e] = x1[e]; }
            if (MODE != 0) {
                const f4 g0 = *reinterpret_cast<const f4*>(gain + col), g1 = *reinterpret_cast<const f4*>(gain + col + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = (v[e] * inv) * g0[e]; v[4 + e] = (v[4 + e] * inv) * g1[e]; }
            }
        }
        ximg_store8(img, MT, tok, col, v);
    }
}

#ifdef THK_ATTN_TRACE      // the same development build also stamps the reducers (tools/dev/attn_trace.py reducers): [kernel 0..2][workgroup][stamp]
__device__ unsigned long long g_rd_trace[3 * 4096 * 4];
#define RD_T(k, i) { if (threadIdx.x == 0) g_rd_trace[(((k) * 4096 + ((blockIdx.y * gridDim.x + blockIdx.x) & 4095)) * 4) + (i)] = wall_clock64(); }
extern "C" __attribute__((visibility("default"))) int thk_debug_reduce_trace(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rd_trace), sizeof(unsigned long long) * 3 * 4096 * 4);
}
#else
#define RD_T(k, i)
#endif
// ---- reducers ----------------------------------------------------------------------------------
// A partial slot holds one 256-row x Mpad-token tile in the MFMA accumulator's own order ("fragment order"), so the
// GEMM's spill is a sequence of contiguous 1 KiB stores and the reducers' reads are contiguous too: float4 number
// q = (((wave*NF + f)*MT + t)*4 + g)*64 + lane  holds rows [(wave*NF + f)*32 + 8g + 4*(lane>>5), +4) of token t*32 + (lane&31)
// (NF = tile_rows / 128 fragments per wave).
// (Token-major slots made every spill instruction touch 32 different lines: the spill cost 58 us per layer.)
struct FragPos { int tok, row; };
__device__ __forceinline__ FragPos frag_decode(int q, int MT) {
    const int lane = q & 63, g = (q >> 6) & 3, rest = q >> 8, t = rest % MT, wf = rest / MT;
    return FragPos{t * 32 + (lane & 31), wf * 32 + 8 * g + 4 * (lane >> 5)};
}
// sum over the workgroups that worked on row-block rbk, in workgroup order (deterministic)
__device__ __forceinline__ f4 sum_partials(const float* __restrict__ part, const PrefillPlan& p, int rbk, int q) {
    const int lo = rbk * p.nchunks, hi = lo + p.nchunks - 1;
    const int b0 = lo / p.per, b1 = hi / p.per;
    // loads four slots ahead of the adds; the sum itself stays strictly in workgroup order
    const float* base = part + (size_t)q * 4;
    auto slot_of = [&](int b) { return base + ((size_t)b * p.maxseg + (rbk - (b * p.per) / p.nchunks)) * p.slot_floats; };
    f4 s = *reinterpret_cast<const f4*>(slot_of(b0));
    int b = b0 + 1;
    for (; b + 3 <= b1; b += 4) {
        const f4 v0 = *reinterpret_cast<const f4*>(slot_of(b)), v1 = *reinterpret_cast<const f4*>(slot_of(b + 1));
        const f4 v2 = *reinterpret_cast<const f4*>(slot_of(b + 2)), v3 = *reinterpret_cast<const f4*>(slot_of(b + 3));
        s = s + v0; s = s + v1; s = s + v2; s = s + v3;
    }
    for (; b <= b1; ++b) s = s + *reinterpret_cast<const f4*>(slot_of(b));
    return s;
}
// one thread per float4 of every tile: grid = row-blocks x (MT * 2048)
// Y[tok][r] = sum  (mode 0)   |   Y[tok][r] += sum  (mode 1: residual)
__global__ __launch_bounds__(256) void reduce_store_kernel(const float* __restrict__ part, PrefillPlan p, float* __restrict__ Y, int mode) {
    const int q = blockIdx.x * 256 + threadIdx.x, rbk = blockIdx.y;
    const FragPos fp = frag_decode(q, p.MT);
    const int r = rbk * p.tile_rows + fp.row;
    if (fp.tok >= p.M || r >= p.R) return;
    f4 s = sum_partials(part, p, rbk, q);
    f4* dst = reinterpret_cast<f4*>(Y + (size_t)fp.tok * p.R + r);
    if (mode == 1) s = *dst + s;
    *dst = s;
}
// x = X + sum (residual, K11) -> X, and straight on to the NEXT GEMM's operand: the hi/lo image of x * gain (deferred norm, see
// ximg_from_rows_kernel) and this workgroup's share of every token's sum of squares.  A workgroup = 256 consecutive q = one
// (token tile, 32-row fragment): 32 tokens x 8 threads, each with 4 rows.  Replaces reduce_store + norm -> image (two launches).
__global__ __launch_bounds__(256) void reduce_resid_ximg_kernel(const float* __restrict__ part, PrefillPlan p, float* __restrict__ X, const float* __restrict__ gain,
                                                                char* __restrict__ img, unsigned long long* __restrict__ ssq, const unsigned long long* __restrict__ ssq_scale) {
    __shared__ float red[4][32];
    RD_T(0, 0)
    const int q = blockIdx.x * 256 + threadIdx.x, rbk = blockIdx.y;
    const FragPos fp = frag_decode(q, p.MT);
    const int r = rbk * p.tile_rows + fp.row, tok = fp.tok;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 hi = h4{0, 0, 0, 0}, lo = h4{0, 0, 0, 0};
    float ss = 0.f;
    const bool live = tok < p.M && r < p.R;
    if (live) {
        f4* dst = reinterpret_cast<f4*>(X + (size_t)tok * p.R + r);
        const f4 x = *dst + sum_partials(part, p, rbk, q);
        RD_T(0, 1)
        *dst = x;
        const f4 g = *reinterpret_cast<const f4*>(gain + r);
        const float sc = ssq_pow2(ssq_scale, tok, p.R);          // power of two near this token's 1/rms (from its previous norm input)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = (x[e] * sc) * g[e];
            hi[e] = (_Float16)v; lo[e] = (_Float16)(v - (float)hi[e]);
            ss += x[e] * x[e];
        }
    }
    if (r < p.R) {                                  // pad tokens of the image are zeros
        const size_t sub = (size_t)(r & 4) * 2;     // second half of the 16-byte piece
        *reinterpret_cast<h4*>(img + ximg_off(p.MT, 0, tok, r) + sub) = hi;
        *reinterpret_cast<h4*>(img + ximg_off(p.MT, 1, tok, r) + sub) = lo;
    }
    // the token's 32 rows of this workgroup: lanes l, l + 32 (row halves), then the four waves (g)
    RD_T(0, 2)
    ss += __shfl_xor(ss, 32, 64);
    if ((threadIdx.x & 63) < 32) red[threadIdx.x >> 6][threadIdx.x & 31] = ss;
    __syncthreads();
    if (threadIdx.x < 32) {
        const int t2 = fp.tok;                      // thread l < 32: lane l of wave 0 -> token (tile) * 32 + l
        if (t2 < p.M) ssq_add(ssq, t2, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
    }
    RD_T(0, 3)
}
// q -> RoPE -> Q[tok];  k -> RoPE -> K-cache row n_past+tok;  v -> V-cache row  (K6, th-llama.cpp:318-339)
// ssq != NULL: the GEMM ran on the un-normalised image (deferred norm): the token's 1/rms is applied here
template <bool KVH>
__global__ __launch_bounds__(256) void reduce_qkv_kernel(const float* __restrict__ part, PrefillPlan p, const float* __restrict__ tab, int n_past, int D,
                                                         float* __restrict__ Q, void* __restrict__ kc, void* __restrict__ vc, const unsigned long long* __restrict__ ssq,
                                                         const unsigned long long* __restrict__ ssq_scale) {
    const int q = blockIdx.x * 256 + threadIdx.x, rbk = blockIdx.y;
    const FragPos fp = frag_decode(q, p.MT);
    const int mat = rbk / p.rb_per_mat, r = (rbk % p.rb_per_mat) * p.tile_rows + fp.row, tok = fp.tok;
    RD_T(1, 0)
    if (tok >= p.M || r >= p.R) return;
    f4 s = sum_partials(part, p, rbk, q);
    RD_T(1, 1)
    if (ssq) s = s * (ssq_inv(ssq, tok, p.C) / ssq_pow2(ssq_scale, tok, p.C));      // the division by a power of two is exact
    RD_T(1, 2)
    const int pos = n_past + tok;
    if (mat < 2) {
        const int half = D >> 1, jp = (r % D) >> 1;
        const f4 cs = *reinterpret_cast<const f4*>(tab + ((size_t)pos * half + jp) * 2);     // cos0 sin0 cos1 sin1
        s = f4{s[0] * cs[0] - s[1] * cs[1], s[0] * cs[1] + s[1] * cs[0], s[2] * cs[2] - s[3] * cs[3], s[2] * cs[3] + s[3] * cs[2]};
    }
    if (mat == 0) { *reinterpret_cast<f4*>(Q + (size_t)tok * p.R + r) = s; return; }
    if (KVH) {          // optional f16 cache: the same RNE rounding as the decode path's append
        typedef _Float16 h4c __attribute__((ext_vector_type(4)));
        *reinterpret_cast<h4c*>(reinterpret_cast<_Float16*>(mat == 1 ? kc : vc) + (size_t)pos * p.R + r) = h4c{(_Float16)s[0], (_Float16)s[1], (_Float16)s[2], (_Float16)s[3]};
    } else {
        *reinterpret_cast<f4*>(reinterpret_cast<float*>(mat == 1 ? kc : vc) + (size_t)pos * p.R + r) = s;
    }
}
// hidden = silu(w1 x) * (w3 x)  (K10, K11) written straight into the X image of the w2 GEMM (C = R of this plan);
// a thread owns 4 columns = half of a 16-byte image piece
__global__ __launch_bounds__(256) void reduce_swiglu_ximg_kernel(const float* __restrict__ part, PrefillPlan p, char* __restrict__ img, const unsigned long long* __restrict__ ssq,
                                                                 const unsigned long long* __restrict__ ssq_scale) {
    const int q = blockIdx.x * 256 + threadIdx.x, rbk = blockIdx.y;     // rbk < rb_per_mat: w1's tile; w3's is rbk + rb_per_mat
    const FragPos fp = frag_decode(q, p.MT);
    const int r = rbk * p.tile_rows + fp.row, tok = fp.tok;
    if (r >= p.R) return;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 hi = h4{0, 0, 0, 0}, lo = h4{0, 0, 0, 0};
    RD_T(2, 0)
    if (tok < p.M) {
        f4 u1 = sum_partials(part, p, rbk, q), u3 = sum_partials(part, p, rbk + p.rb_per_mat, q);
        RD_T(2, 1)
        if (ssq) { const float inv = ssq_inv(ssq, tok, p.C) / ssq_pow2(ssq_scale, tok, p.C); u1 = u1 * inv; u3 = u3 * inv; }    // deferred norm
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sl = u1[e] / (1.0f + expf(-u1[e])), v = sl * u3[e];
            hi[e] = (_Float16)v; lo[e] = (_Float16)(v - (float)hi[e]);
        }
    }
    const size_t sub = (size_t)(r & 4) * 2;                              // second half of the piece
    *reinterpret_cast<h4*>(img + ximg_off(p.MT, 0, tok, r) + sub) = hi;
    *reinterpret_cast<h4*>(img + ximg_off(p.MT, 1, tok, r) + sub) = lo;
    RD_T(2, 3)
}

#ifdef THK_PREFILL_TRACE
extern "C" __attribute__((visibility("default"))) int thk_debug_prefill_trace(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pf_trace), sizeof(unsigned long long) * 4 * 256 * 4 * 12);
}
#endif
hipError_t launch_prefill_ximg(const float* X, const float* gain, int M, int C, void* ximg, hipStream_t st, unsigned long long* ssq) {
    const int MT = prefill_token_tiles(M);
    if (M < 1 || M > 256 || C % kKC != 0 || (ssq && !gain)) return hipErrorInvalidValue;
    if (gain && ssq) hipLaunchKernelGGL(ximg_from_rows_kernel<2>, dim3(MT * 32), dim3(256), 0, st, X, gain, M, MT, C, (char*)ximg, ssq);
    else if (gain) hipLaunchKernelGGL(ximg_from_rows_kernel<1>, dim3(MT * 32), dim3(256), 0, st, X, gain, M, MT, C, (char*)ximg, ssq);
    else hipLaunchKernelGGL(ximg_from_rows_kernel<0>, dim3(MT * 32), dim3(256), 0, st, X, gain, M, MT, C, (char*)ximg, ssq);
    return hipGetLastError();
}
hipError_t launch_prefill_reduce_resid_ximg(const float* part, const PrefillPlan& p, float* X, const float* gain, void* ximg, unsigned long long* ssq,
                                            const unsigned long long* ssq_scale, hipStream_t st) {
    if (!gain || !ximg || !ssq || !ssq_scale || p.R % kKC != 0) return hipErrorInvalidValue;      // the output row count is the next GEMM's column count
    hipLaunchKernelGGL(reduce_resid_ximg_kernel, dim3(p.MT * p.tile_rows / 32, p.rb_total), dim3(256), 0, st, part, p, X, gain, (char*)ximg, ssq, ssq_scale);
    return hipGetLastError();
}
hipError_t launch_prefill_gemm(const uint16_t* const* W, const PrefillPlan& plan_in, const void* ximg, float* part, hipStream_t st) {
    PrefillPlan p = plan_in;
#ifdef THK_PREFILL_TRACE
    { static int n_launch = 0; p.packed = (p.packed & 1) | ((n_launch++ & 3) << 8); }
#endif
    if (p.M < 1 || p.M > 256 || p.MT != prefill_token_tiles(p.M) || p.C % kKC != 0 || p.R % 4 != 0 || p.nmat < 1 || p.nmat > 3) return hipErrorInvalidValue;
    const _Float16* w0 = reinterpret_cast<const _Float16*>(W[0]);
    const _Float16* w1 = reinterpret_cast<const _Float16*>(W[p.nmat > 1 ? 1 : 0]);
    const _Float16* w2 = reinterpret_cast<const _Float16*>(W[p.nmat > 2 ? 2 : 0]);
    hipError_t e = hipSuccess;
#define THK_V3K(MTV, NFV, PKV)                                                                                           \
    {                                                                                                                    \
        const size_t lds = (ximg_stage_bytes(MTV) + (size_t)NFV * 8192) * kNST + 4096;   /* + 1 KiB per wave for dummy loads */ \
        static bool attr_done[kMaxDevices] = {};     /* the attribute is per device */                                   \
        const int dev = current_device();                                                                                \
        if (!attr_done[dev]) { e = hipFuncSetAttribute((const void*)gemm_prefill_v3_kernel<MTV, NFV, PKV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done[dev] = (e == hipSuccess); } \
        if (e == hipSuccess) hipLaunchKernelGGL((gemm_prefill_v3_kernel<MTV, NFV, PKV>), dim3(p.G), dim3(256), lds, st, w0, w1, w2, (const char*)ximg, \
                                                p.nchunks, p.per, p.rb_per_mat, p.rb_total, p.R, p.C, part, p); \
    }
#define THK_V3(MTV, NFV) { if (p.packed & 1) THK_V3K(MTV, NFV, true) else THK_V3K(MTV, NFV, false) }
#define THK_V3HK(NFV, PKV)                                                                                               \
    {                                                                                                                    \
        const size_t lds = (ximg_stage_bytes(8) + (size_t)NFV * 8192) * kNSTH + 4096;                                    \
        static bool attr_done[kMaxDevices] = {};                                                                         \
        const int dev = current_device();                                                                                \
        if (!attr_done[dev]) { e = hipFuncSetAttribute((const void*)gemm_prefill_v3h_kernel<NFV, PKV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done[dev] = (e == hipSuccess); } \
        if (e == hipSuccess) hipLaunchKernelGGL((gemm_prefill_v3h_kernel<NFV, PKV>), dim3(p.G), dim3(256), lds, st, w0, w1, w2, (const char*)ximg, \
                                                p.nchunks, p.per, p.rb_per_mat, p.rb_total, p.R, p.C, part, p);         \
    }
#define THK_V3GK(NFV, PKV)                                                                                               \
    {                                                                                                                    \
        const size_t lds = (ximg_stage_bytes(8) + (size_t)NFV * 8192) * kNSTH + 4096;                                    \
        static bool attr_done[kMaxDevices] = {};                                                                         \
        const int dev = current_device();                                                                                \
        if (!attr_done[dev]) { e = hipFuncSetAttribute((const void*)gemm_prefill_v3g_kernel<NFV, PKV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done[dev] = (e == hipSuccess); } \
        if (e == hipSuccess) hipLaunchKernelGGL((gemm_prefill_v3g_kernel<NFV, PKV>), dim3(p.G), dim3(256), lds, st, w0, w1, w2, (const char*)ximg, \
                                                p.nchunks, p.per, p.rb_per_mat, p.rb_total, p.R, p.C, part, p);         \
    }
    if (p.MT == 8) {                       // 129 .. 256 tokens: the two-half kernels (packed bit 1: the 2 x 2 wave grid, round 6)
        if (p.packed & 2) {
            if (p.tile_rows == 128) { if (p.packed & 1) THK_V3GK(1, true) else THK_V3GK(1, false) }
            else { if (p.packed & 1) THK_V3GK(2, true) else THK_V3GK(2, false) }
        } else {
            if (p.tile_rows == 128) { if (p.packed & 1) THK_V3HK(1, true) else THK_V3HK(1, false) }
            else { if (p.packed & 1) THK_V3HK(2, true) else THK_V3HK(2, false) }
        }
        return e != hipSuccess ? e : hipGetLastError();
    }
    if (p.tile_rows == 128) switch (p.MT) {
        case 1: THK_V3(1, 1) break;
        case 2: THK_V3(2, 1) break;
        case 3: THK_V3(3, 1) break;
        default: THK_V3(4, 1) break;
    } else switch (p.MT) {
        case 1: THK_V3(1, 2) break;
        case 2: THK_V3(2, 2) break;
        case 3: THK_V3(3, 2) break;
        default: THK_V3(4, 2) break;
    }
#undef THK_V3
#undef THK_V3K
#undef THK_V3HK
#undef THK_V3GK
    return e != hipSuccess ? e : hipGetLastError();
}
hipError_t launch_prefill_reduce_store(const float* part, const PrefillPlan& p, float* Y, bool residual, hipStream_t st) {
    hipLaunchKernelGGL(reduce_store_kernel, dim3(p.MT * p.tile_rows / 32, p.rb_total), dim3(256), 0, st, part, p, Y, residual ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_prefill_reduce_qkv(const float* part, const PrefillPlan& p, const float* rope_tab, int n_past, int D, float* Q, void* kcache, void* vcache, bool kv_f16, hipStream_t st,
                                     const unsigned long long* ssq, const unsigned long long* ssq_scale) {
    if (ssq && !ssq_scale) return hipErrorInvalidValue;
    if (kv_f16) hipLaunchKernelGGL(reduce_qkv_kernel<true>, dim3(p.MT * p.tile_rows / 32, p.rb_total), dim3(256), 0, st, part, p, rope_tab, n_past, D, Q, kcache, vcache, ssq, ssq_scale);
    else hipLaunchKernelGGL(reduce_qkv_kernel<false>, dim3(p.MT * p.tile_rows / 32, p.rb_total), dim3(256), 0, st, part, p, rope_tab, n_past, D, Q, kcache, vcache, ssq, ssq_scale);
    return hipGetLastError();
}
hipError_t launch_prefill_reduce_swiglu(const float* part, const PrefillPlan& p, void* ximg_out, hipStream_t st, const unsigned long long* ssq, const unsigned long long* ssq_scale) {
    if (ssq && !ssq_scale) return hipErrorInvalidValue;
    hipLaunchKernelGGL(reduce_swiglu_ximg_kernel, dim3(p.MT * p.tile_rows / 32, p.rb_per_mat), dim3(256), 0, st, part, p, (char*)ximg_out, ssq, ssq_scale);
    return hipGetLastError();
}

// ---------------------------------------------------------------- causal prefill attention on MFMA
// out[q, h, :] = softmax_{pos <= n_past + q}( Q[q,h,:] . K[pos,h,:] / sqrt(D) ) V[pos,h,:]   for the M queries of a slab.
// Workgroup = (head, 32-query tile), 4 waves; wave w takes the 32-position tiles w, w+4, ... with its own online
// softmax state, the four states are merged through LDS at the end.  Per tile:
//   S^T[pos][q] = K Q^T      A = K tile rows (f32 -> hi/lo f16, staged in LDS: coalesced global reads, XOR-swizzled
//                            rows => conflict-free fragment reads), B = Q^T fragments held in registers;
//   online softmax           with S TRANSPOSED a lane owns ONE query (column) and 16 of the tile's 32 positions, so
//                            the row max / row sum are 16 register ops + one xor-32 shuffle, and the O^T accumulator
//                            (columns = the same query) is rescaled by a per-lane scalar;
//   O^T[d][q] += V^T P^T     B = P^T straight from the S^T registers (k-slot order = register order; the A side
//                            gathers V[pos][d] from the LDS tile in that same order with 16-bit reads).
// Every product of two f32 operands is three f16 MFMAs (hi.hi + hi.lo + lo.hi): f32-class accuracy, as in the GEMM.
// Replaces one workgroup per (head, query) re-reading that head's K/V from L2: 17 -> 105 us per slab-layer as the
// context grew from 128 to 512 positions.
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
#ifdef THK_ATTN_TRACE      // development build only (tools/dev/attn_trace.py): 100 MHz wall-clock stamps of one wave per workgroup at the kernel's phase edges
__device__ unsigned long long g_at_trace[1024 * 4 * 12];
#define AT_T(i) { if (lane == 0) g_at_trace[((blockIdx.x & 1023) * 4 + wave) * 12 + (i)] = wall_clock64(); }
extern "C" __attribute__((visibility("default"))) int thk_debug_attn_trace(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_at_trace), sizeof(unsigned long long) * 1024 * 4 * 12);
}
#else
#define AT_T(i)
#endif
__device__ __forceinline__ void split4(const f4 v, h4v& hi, h4v& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { hi[e] = (_Float16)v[e]; lo[e] = (_Float16)(v[e] - (float)hi[e]); }
}
// A 32 x D f32 tile of the cache -> registers (all loads issued back to back: written as load+convert+store per
// segment, hipcc waits vmcnt(0) after every load, 32 serialized L2 round trips per tile = 17 us) ...
template <int D>
struct AttnTileRegs { f4 v[32 * (D / 4) / 64]; };
template <int D, bool KVH>
__device__ __forceinline__ void attn_load_tile(AttnTileRegs<D>& r, const void* __restrict__ cache, int p0, int p_last, int E, int hcol, int lane) {
    constexpr int SPR = D / 4;                          // float4 segments per row
#pragma unroll
    for (int i = 0; i < 32 * SPR / 64; ++i) {
        const int idx = i * 64 + lane, row = idx / SPR, seg = idx % SPR;
        const int prow = p0 + row < p_last ? p0 + row : p_last;          // rows past the context repeat the last cached row (masked later)
        if (KVH) {
            const h4v h = *reinterpret_cast<const h4v*>(reinterpret_cast<const _Float16*>(cache) + (size_t)prow * E + hcol + seg * 4);
            r.v[i] = f4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        } else {
            r.v[i] = *reinterpret_cast<const f4*>(reinterpret_cast<const float*>(cache) + (size_t)prow * E + hcol + seg * 4);
        }
    }
}
// ... -> hi/lo f16 in the wave's LDS tile, rows of D halfs; SWZ: 16-byte pieces XOR-swizzled by the row (fragment reads)
template <int D, bool SWZ>
__device__ __forceinline__ void attn_store_tile(const AttnTileRegs<D>& r, int lane, _Float16* t_hi, _Float16* t_lo) {
    constexpr int SPR = D / 4;
#pragma unroll
    for (int i = 0; i < 32 * SPR / 64; ++i) {
        const int idx = i * 64 + lane, row = idx / SPR, seg = idx % SPR;
        h4v hi, lo; split4(r.v[i], hi, lo);
        const int piece = seg >> 1, pos = SWZ ? (piece ^ (row & (D / 8 - 1))) : piece;
        const int off = row * D + pos * 8 + (seg & 1) * 4;
        *reinterpret_cast<h4v*>(t_hi + off) = hi;
        *reinterpret_cast<h4v*>(t_lo + off) = lo;
    }
}
// IMG: instead of out[M, E] f32, write the hi/lo X image of the wo GEMM directly (one launch and one 2 MB round trip less)
template <int D, bool IMG, bool KVH>
__global__ __launch_bounds__(256) void attn_prefill_mfma_kernel(const float* __restrict__ Q, const void* __restrict__ Kc, const void* __restrict__ Vc,
                                                                int n_past, int M, int H, float scale, float* __restrict__ out, char* __restrict__ img,
                                                                int img_MT, int img_tok0 /* IMG: token tiles of the image, and the image row of query 0 */) {
    constexpr int KS = D / 16, DB = D / 32;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];      // 4 waves x {hi, lo} x 32 x D halfs; reused for the merge
    __shared__ float sm_m[4][32], sm_l[4][32], sm_f[4][32];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = blockIdx.x % H, q0 = (blockIdx.x / H) * 32;
    const int E = H * D, hcol = h * D;
    const int qn = lane & 31, half = lane >> 5;
    _Float16* t_hi = reinterpret_cast<_Float16*>(lds_raw) + (size_t)wave * 2 * 32 * D;
    _Float16* t_lo = t_hi + 32 * D;

    AT_T(0)
    h8 qh[KS], ql[KS];                                  // B fragments of Q^T: lane (q, half) holds d = 16 ks + 8 half + e
    {
        const int qrow = q0 + qn < M ? q0 + qn : M - 1;
        const float* qp = Q + (size_t)qrow * E + hcol + half * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f4 a = *reinterpret_cast<const f4*>(qp + ks * 16), b = *reinterpret_cast<const f4*>(qp + ks * 16 + 4);
            h4v ah, al, bh, bl; split4(a, ah, al); split4(b, bh, bl);
#pragma unroll
            for (int e = 0; e < 4; ++e) { qh[ks][e] = ah[e]; qh[ks][4 + e] = bh[e]; ql[ks][e] = al[e]; ql[ks][4 + e] = bl[e]; }
        }
    }
    float m = -INFINITY, l = 0.f;
    f16v o[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[b][i] = 0.f;

    const int q_last = q0 + 31 < M ? q0 + 31 : M - 1;
    const int p_last = n_past + M - 1;                  // last cached row
    const int ntiles = (n_past + q_last) / 32 + 1;      // tiles that hold a position some query of this tile may see
    const int qabs = n_past + q0 + qn;                  // this lane's query position
    // The K tile of a wave's NEXT round is requested under the current round's P V phase (the registers are free once V is in LDS), the first one next to
    // the Q rows above: a wave with several tiles (a slab behind cached rows, the later query tiles of a 256-token slab) no longer pays a memory round trip
    // per tile.  Branch-free: past the last tile the request repeats it (a load behind a branch is waited for at the join).
    AT_T(1)
    AttnTileRegs<D> tr;
    attn_load_tile<D, KVH>(tr, Kc, (wave < ntiles ? wave : ntiles - 1) * 32, p_last, E, hcol, lane);
    for (int t = wave; t < ntiles; t += 4) {
        const int p0 = t * 32;
        __builtin_amdgcn_sched_barrier(0);
        attn_store_tile<D, true>(tr, lane, t_hi, t_lo);
        AT_T(2)
        __builtin_amdgcn_sched_barrier(0);
        attn_load_tile<D, KVH>(tr, Vc, p0, p_last, E, hcol, lane);           // V's round trip runs under the S^T / softmax phase
        __builtin_amdgcn_sched_barrier(0);
        f16v s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {               // A fragment: row = position (lane & 31), piece 2 ks + half
            const int off = qn * D + (((ks * 2 + half) ^ (qn & (D / 8 - 1))) * 8);
            const h8 kh = *reinterpret_cast<const h8*>(t_hi + off), kl = *reinterpret_cast<const h8*>(t_lo + off);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s, 0, 0, 0);
        }
        AT_T(3)
        // s[r] = S[pos = p0 + (r&3) + 8 (r>>2) + 4 half][this lane's query]; scale after the sum (th.cpp:527-529), causal mask
        float bm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pos = p0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            s[r] = pos <= qabs ? s[r] * scale : -INFINITY;
            bm = fmaxf(bm, s[r]);
        }
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float mn = fmaxf(m, bm);
        const float alpha = (m == -INFINITY) ? 0.f : expf(m - mn);       // mn == -inf only while m == -inf
        float ps = 0.f;
        h8 ph[2], pl[2];                                // B fragments of P^T: k-slot (s2, half, e) = register 8 s2 + e
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = (s[r] == -INFINITY) ? 0.f : expf(s[r] - mn);
            ps += pv;
            const _Float16 hi = (_Float16)pv;
            ph[r >> 3][r & 7] = hi; pl[r >> 3][r & 7] = (_Float16)(pv - (float)hi);
        }
        ps += __shfl_xor(ps, 32, 64);
        l = l * alpha + ps; m = mn;
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[b][i] *= alpha;
        __builtin_amdgcn_sched_barrier(0);
        AT_T(4)
        attn_store_tile<D, false>(tr, lane, t_hi, t_lo);                 // same LDS region: K is consumed
        AT_T(5)
        __builtin_amdgcn_sched_barrier(0);
        attn_load_tile<D, KVH>(tr, Kc, (t + 4 < ntiles ? t + 4 : ntiles - 1) * 32, p_last, E, hcol, lane);     // next round's K, in flight under P V
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {            // A fragment of V^T: row d = 32 b + (lane & 31), k-slot e -> position 16 s2 + (e&3) + 8 (e>>2) + 4 half
                h8 vh, vl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int off = (16 * s2 + (e & 3) + 8 * (e >> 2) + 4 * half) * D + 32 * b + qn;
                    vh[e] = t_hi[off]; vl[e] = t_lo[off];
                }
                o[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[s2], o[b], 0, 0, 0);
                o[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[s2], o[b], 0, 0, 0);
                o[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[s2], o[b], 0, 0, 0);
            }
    }
    AT_T(6)
    // merge the four waves: o[b][r] = O^T[d = 32 b + (r&3) + 8 (r>>2) + 4 half][q = qn]
    __syncthreads();                                    // every wave is done with its tile region
    AT_T(7)
    float* sm_o = reinterpret_cast<float*>(lds_raw);    // [wave][d][32]
    if (half == 0) { sm_m[wave][qn] = m; sm_l[wave][qn] = l; }
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) sm_o[((size_t)wave * D + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half) * 32 + qn] = o[b][r];
    __syncthreads();
    if (threadIdx.x < 128) {                            // per (wave, query) weight exp(m_w - max m) / denominator, once
        const int w = threadIdx.x >> 5, q = threadIdx.x & 31;
        const float mm = fmaxf(fmaxf(sm_m[0][q], sm_m[1][q]), fmaxf(sm_m[2][q], sm_m[3][q]));
        float den = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) den += (sm_m[u][q] == -INFINITY) ? 0.f : sm_l[u][q] * expf(sm_m[u][q] - mm);
        sm_f[w][q] = (sm_m[w][q] == -INFINITY) ? 0.f : expf(sm_m[w][q] - mm) / den;
    }
    __syncthreads();
    AT_T(8)
    // A thread finishes 8 consecutive columns of one query = one 16-byte piece of the image (hi) and one of its lo half - or two float4 of `out`.  (Round 5:
    // the first version stored the image's halfs one by one, 32 two-byte stores per thread: 7.3 of the launch's 16 us, tools/dev/attn_trace.py.)
    for (int idx = threadIdx.x; idx < 32 * (D / 8); idx += 256) {
        const int q = idx & 31, d0 = (idx >> 5) * 8;                    // consecutive lanes = consecutive queries: conflict-free LDS reads
        if (!IMG && q0 + q >= M) continue;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) a += sm_o[((size_t)w * D + d0 + e) * 32 + q] * sm_f[w][q];
            acc[e] = a;
        }
        if (IMG) {                                      // pad rows (tok >= M) of the image are zeros
            h8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = q0 + q >= M ? 0.f : acc[e];
                hi[e] = (_Float16)a; lo[e] = (_Float16)(a - (float)hi[e]);
            }
            const int col = hcol + d0;
            *reinterpret_cast<h8*>(img + ximg_off(img_MT, 0, img_tok0 + q0 + q, col)) = hi;
            *reinterpret_cast<h8*>(img + ximg_off(img_MT, 1, img_tok0 + q0 + q, col)) = lo;
        } else {
            float* dst = out + (size_t)(q0 + q) * E + hcol + d0;
            *reinterpret_cast<f4*>(dst) = f4{acc[0], acc[1], acc[2], acc[3]};
            *reinterpret_cast<f4*>(dst + 4) = f4{acc[4], acc[5], acc[6], acc[7]};
        }
    }
    AT_T(9)
}

template <int D, bool IMG, bool KVH>
static hipError_t launch_attn_prefill_t(const float* Q, const void* Kc, const void* Vc, int n_past, int M, int H, float* out, char* img, hipStream_t st, int img_MT, int img_tok0, int q_tiles) {
    const int grid = H * (q_tiles > 0 ? q_tiles : (M + 31) / 32);     // q_tiles > the queries' own tiles: the extra workgroups write the image's zero rows
    const size_t lds = (size_t)4 * 2 * 32 * D * 2;
    hipError_t e = hipSuccess;
    static bool done[kMaxDevices] = {};                 // the attribute is per device (64 KB dynamic + the static arrays)
    const int dev = current_device();
    if (!done[dev]) { e = hipFuncSetAttribute((const void*)attn_prefill_mfma_kernel<D, IMG, KVH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done[dev] = (e == hipSuccess); }
    if (e == hipSuccess) hipLaunchKernelGGL((attn_prefill_mfma_kernel<D, IMG, KVH>), dim3(grid), dim3(256), lds, st, Q, Kc, Vc, n_past, M, H, 1.0f / sqrtf((float)D), out, img, img_MT > 0 ? img_MT : (M + 31) / 32, img_tok0);
    return e != hipSuccess ? e : hipGetLastError();
}
template <int D, bool KVH>
static hipError_t launch_attn_prefill_d(const float* Q, const void* Kc, const void* Vc, int n_past, int M, int H, float* out, void* ximg, hipStream_t st, int img_MT, int img_tok0, int q_tiles) {
    return out ? launch_attn_prefill_t<D, false, KVH>(Q, Kc, Vc, n_past, M, H, out, nullptr, st, 0, 0, 0) : launch_attn_prefill_t<D, true, KVH>(Q, Kc, Vc, n_past, M, H, nullptr, (char*)ximg, st, img_MT, img_tok0, q_tiles);
}
// out != null: out[M, H*D] f32;  otherwise ximg = the hi/lo X image (C = H*D, M <= 128) of the GEMM that consumes the attention output.
// kv_f16: the caches hold binary16 rows (optional f16 KV cache); the tiles are widened to f32 as they are loaded.
hipError_t launch_attn_prefill_mfma(const float* Q, const void* Kc, const void* Vc, bool kv_f16, int n_past, int M, int H, int D, float* out, void* ximg, hipStream_t st,
                                    int img_MT, int img_tok0, int q_tiles) {
    if ((D != 64 && D != 128) || (!out && (!ximg || M > 256 || (H * D) % kKC != 0))) return hipErrorInvalidValue;
    if (img_MT < 0 || img_MT > 8 || img_tok0 < 0 || q_tiles < 0 || q_tiles > 8 || (!out && M > 128 && img_MT != 8)) return hipErrorInvalidValue;
    if (D == 128) return kv_f16 ? launch_attn_prefill_d<128, true>(Q, Kc, Vc, n_past, M, H, out, ximg, st, img_MT, img_tok0, q_tiles) : launch_attn_prefill_d<128, false>(Q, Kc, Vc, n_past, M, H, out, ximg, st, img_MT, img_tok0, q_tiles);
    return kv_f16 ? launch_attn_prefill_d<64, true>(Q, Kc, Vc, n_past, M, H, out, ximg, st, img_MT, img_tok0, q_tiles) : launch_attn_prefill_d<64, false>(Q, Kc, Vc, n_past, M, H, out, ximg, st, img_MT, img_tok0, q_tiles);
}

size_t gemm_prefill_workspace_bytes(int M, int R, int C) {
    const PrefillPlan p = prefill_plan(M < 128 ? M : 128, R, 1, C, 0, 0);
    return (p.ximg_bytes + 255) / 256 * 256 + p.part_floats * 4;
}

// Y[M,R] = X[M,C] * W[R,C]^T for one matrix (the thk_gemm_f16_prefill operator): image -> GEMM -> reduce, 128 tokens at a time
hipError_t launch_gemm_f16_prefill(const uint16_t* W, int R, int C, const float* X, int M, float* Y, void* workspace, hipStream_t st) {
    if (C % kKC != 0 || R % 4 != 0) return hipErrorInvalidValue;
    for (int m0 = 0; m0 < M; m0 += 128) {
        const int mc = (M - m0) < 128 ? (M - m0) : 128;
        const PrefillPlan p = prefill_plan(mc, R, 1, C, 0, 0);
        char* img = reinterpret_cast<char*>(workspace);
        float* part = reinterpret_cast<float*>(img + (p.ximg_bytes + 255) / 256 * 256);
        hipError_t e = launch_prefill_ximg(X + (size_t)m0 * C, nullptr, mc, C, img, st);
        if (e == hipSuccess) e = launch_prefill_gemm(&W, p, img, part, st);
        if (e == hipSuccess) e = launch_prefill_reduce_store(part, p, Y + (size_t)m0 * R, false, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace thk
