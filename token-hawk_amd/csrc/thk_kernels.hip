// thk_kernels.hip — hand-written CDNA4 (gfx950) kernels for the single-token
// LLaMA decode path.  wave = 64 lanes everywhere; blocks are 256 threads (4 waves).
//
// Design:
//   * The f16-weight x f32-activation mat-vec is >98 % of the bytes and is pure
//     HBM streaming: one WAVE owns whole rows, every lane issues 16-byte
//     (8 x f16) non-temporal loads, a wave-instruction covers 1 KiB of one row,
//     NR rows x U chunks are issued back to back before the first use so each
//     wave keeps 8-16 KiB in flight.  The f32 activation vector is staged ONCE
//     per block in LDS (split lo/hi float4 layout => conflict-free
//     ds_read_b128), by a prologue that is fused with whatever produced it
//     (RMSNorm*gain, attention split combine, plain copy).  Accumulation is f32
//     FMA on the hardware-converted f16 (v_cvt_f32_f16 == the reference's
//     bit-trick decode, th.cpp:363-394).  Row sums are reduced with DPP.
//   * Epilogues fuse what the reference runs as separate dispatches: RoPE +
//     K/V append, residual add, SiLU*gate, lm-head split combine + arg-max.
//   * Attention reads the f32 caches in place ([n_ctx,H,D]); no transposed copy.
//   * Positions/tokens are read from device memory so one captured hipGraph
//     serves every token.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "thk_kernels.hpp"
#include "thk_device.hpp"
#include "thk_decode_bodies.hpp"

namespace thk {

// Kernel-argument preload (round 4).  A wave's first instructions used to be s_load_dword of its kernel arguments - a scalar-cache
// miss that goes to L2 and, for the first waves on every XCD, to memory - and nothing, not even the first weight request, could be
// issued before it returned: part of every launch's ~1.8 us ramp, 161 times per token.  gfx950 can deliver the first 14 dwords of
// the kernarg segment in SGPRs at wave launch (hipcc -mllvm -amdgpu-kernarg-preload-count=14, .amdhsa_user_sgpr_kernarg_preload_length;
// the code object keeps a fall-back prologue for firmware without the feature).  Only EXPLICIT scalar arguments are preloaded, not
// the members of a by-value struct, so everything the code needs before its first `s_waitcnt lgkmcnt` - the weight bases, the
// activation and gain vectors, the column / group / row counts and the grid size (gridDim.x is an implicit argument at the END of
// the segment) - travels as leading scalars and the rest of the argument block follows as the struct it always was.
// the argument slot behind the gain pointer: (C | n_groups << 32), or - EPI_ROPE_KV, where both follow from E - the position pointer
// (a specialisation on the plain integer, not std::conditional on `EPI == EPI_ROPE_KV`: the comparison with an unnamed enumerator is
// mangled into the kernel's name, differently by the host and the device pass - "cannot find symbol" at the first launch)
template <int EPI> struct GemvSlotT { typedef unsigned long long type; };
template <> struct GemvSlotT<2> { typedef const int32_t* type; };
static_assert(GEMV_EPI_ROPE_KV == 2, "GemvSlotT<2> is the RoPE epilogue");
template <int EPI> using GemvSlot = typename GemvSlotT<EPI>::type;
template <int NR, int U, int NS, int PRO, int EPI, bool NT, int NSP, bool PIPE, int WPB>
__global__ __launch_bounds__(WPB * 64) void gemv_kernel(const uint16_t* W0, const uint16_t* W1, const uint16_t* W2, const float* x, const float* gain,
                                                        GemvSlot<EPI> cg, int R_or_E, int nblk, const GemvArgs a) {
    GemvArgs b = a;
    b.W[0] = W0; b.W[1] = W1; b.W[2] = W2; b.x = x; b.gain = gain;
    if constexpr (EPI != EPI_ROPE_KV) { b.C = (int)(unsigned)cg; b.n_groups = (int)(unsigned)(cg >> 32); }
    if (EPI == EPI_ROPE_KV) b.E = R_or_E; else b.R = R_or_E;
    // EPI_ROPE_KV (wq | wk | wv are [E, E]: C == E, n_groups == 3 E / NR): the two slots carry the POSITION POINTER instead, so the
    // kernel can request the position with its first instruction - as a struct member its address was a scalar load away, and the
    // prologue's LDS reduction then waited for the position's round trip to memory (same counter), staging the activation vector
    // 2.8 us after entry instead of 1.3 us (tools/step_trace.py)
    if constexpr (EPI == EPI_ROPE_KV) { b.pos_ptr = cg; b.C = R_or_E; b.n_groups = 3 * R_or_E / NR; }
    // PRO_ATTN (wo: one matrix) has no use for the activation / gain / W1 slots: the split partials and the head size travel there, so
    // the prologue's first requests need no scalar load of a struct member (and no wait for one) either
    if (PRO == PRO_ATTN) {
        b.part_o = x; b.part_ml = gain;
        const unsigned long long dm = reinterpret_cast<unsigned long long>(W1);
        b.d_magic = (unsigned)dm; b.D = (int)(dm >> 32);
    }
    gemv_body<NR, U, NS, PRO, EPI, NT, NSP, PIPE, WPB>(b, blockIdx.x, nblk);
}

template <int NR, int U, int NS, int PRO, int EPI, int NSP, bool PIPE, int WPB>
static hipError_t launch_gemv_k(const GemvArgs& a, int grid, bool nt, hipStream_t st) {
    const int ns = NS ? NS : (((a.C >> 3) + 63) >> 6);
    const size_t smem = (size_t)ns * 512 * 4 + 128 + (((EPI == EPI_ROPE_KV || EPI == EPI_SWIGLU) && NR == 1) ? 512 : 0);      // + ysm[32][4] (single-row qkv / w13, gemv_body)
    if ((EPI == EPI_ROPE_KV || EPI == EPI_SWIGLU) && NR == 1 && ((long)grid * WPB * 32 < a.n_groups || (a.n_groups & 1) || WPB != 4)) return hipErrorInvalidValue;     // at most 32 rounds per wave
    (void)nt;   // weights always stream with non-temporal loads (default-policy loads measured 8 % slower)
    auto kn = gemv_kernel<NR, U, NS, PRO, EPI, true, NSP, PIPE, WPB>;
    static size_t attr_set[kMaxDevices] = {};   // per instantiation AND per device; first call happens outside graph capture
    if (smem > 48 * 1024) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
        if (smem > attr_set[dev]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return e;
            attr_set[dev] = smem;
        }
    }
    static const bool dbg = getenv("THK_DEBUG_OCC") != nullptr;      // development: what the runtime says about residency
    if (dbg) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kn), WPB * 64, smem);
        fprintf(stderr, "[thk] gemv<NR=%d,U=%d,NS=%d,PRO=%d,EPI=%d,PIPE=%d,WPB=%d> grid=%d smem=%zu: %d blocks/CU (occupancy API)\n", NR, U, NS, PRO, EPI, (int)PIPE, WPB, grid, smem, nb);
    }
    const uint16_t* w1 = a.W[1];
    if (PRO == PRO_ATTN) {      // see gemv_kernel
        if ((a.D != 64 && a.D != 128 && a.D != 256) || a.C >= 65536) return hipErrorInvalidValue;      // the head sizes attention accepts; ProAttn's thread mapping relies on 64 | D
        const unsigned long long magic = ((1ull << 32) + (unsigned)a.D - 1) / (unsigned)a.D;
        w1 = reinterpret_cast<const uint16_t*>(magic | ((unsigned long long)a.D << 32));
    }
    GemvSlot<EPI> cg;
    if constexpr (EPI == EPI_ROPE_KV) {       // see gemv_kernel
        if (!a.pos_ptr || a.C != a.E || a.n_groups != 3 * a.E / NR) return hipErrorInvalidValue;
        cg = a.pos_ptr;
    } else {
        cg = (unsigned long long)(unsigned)a.C | ((unsigned long long)(unsigned)a.n_groups << 32);
    }
    hipLaunchKernelGGL(kn, dim3(grid), dim3(WPB * 64), smem, st, a.W[0], w1, a.W[2], PRO == PRO_ATTN ? a.part_o : a.x, PRO == PRO_ATTN ? a.part_ml : a.gain, cg,
                       EPI == EPI_ROPE_KV ? a.E : a.R, grid, a);
    return hipGetLastError();
}
template <int NR, int U, int NS, int PRO, int EPI, bool PIPE, int WPB>
static hipError_t launch_gemv_w(const GemvArgs& a, int grid, bool nt, hipStream_t st) {
    if constexpr (PRO == PRO_ATTN) {
        switch (a.nsplit) {
            case 2: return launch_gemv_k<NR, U, NS, PRO, EPI, 2, PIPE, WPB>(a, grid, nt, st);
            case 4: return launch_gemv_k<NR, U, NS, PRO, EPI, 4, PIPE, WPB>(a, grid, nt, st);
            case 8: return launch_gemv_k<NR, U, NS, PRO, EPI, 8, PIPE, WPB>(a, grid, nt, st);
            default: return hipErrorInvalidValue;   // nsplit == 1 uses PRO_COPY on the finished output
        }
    } else {
        return launch_gemv_k<NR, U, NS, PRO, EPI, 0, PIPE, WPB>(a, grid, nt, st);
    }
}
template <int NR, int U, int NS, int PRO, int EPI, bool PIPE>
static hipError_t launch_gemv_t(const GemvArgs& a, int grid, bool nt, hipStream_t st) {
    return launch_gemv_w<NR, U, NS, PRO, EPI, PIPE, kWaves>(a, grid, nt, st);
}

// Slot-count class of a column count: compile-time NS for the LLaMA-7B/13B shapes,
// 0 (run-time loop) for everything else.
static int ns_class(int C) {
    switch (C) { case 4096: return 8; case 5120: return 10; case 11008: return 22; case 13824: return 27; default: return 0; }
}
// (NR rows per wave iteration, U slots per load batch, pipelined loop) variants per class, selectable at run
// time (tunable "gemv_variant_*") so launch geometry can be swept on the GPU without rebuilding.
//   0-2  batch loop: NR*U loads, then their FMAs, per batch (0 = row pairs, 1 = single rows, 2 = row pairs in half batches)
//   5-6  software-pipelined loop (gemv_body PIPE): one whole row group in flight, slot c of the next group requested as soon as
//        slot c of the current one is consumed.  5 = row pair (one row for 11008/13824 columns), 6 = one row
//   3, 4, 7 (four rows per wave in flight) measured slower than their neighbours in rounds 3 and 4 and were retired in round 5:
//        the numbers stay reserved and select 0 / 0 / 5
void gemv_variant(int C, int epi, int nru, int* NR, int* U, int* pipe) {
    const bool pair = false;                   // EPI_ROPE_KV and EPI_SWIGLU take their two rows together (NR = 2) or as single rows (NR = 1, the pair meets in LDS)
    static const int t8[8][3] = {{2, 8, 0}, {1, 8, 0}, {2, 4, 0}, {2, 8, 0}, {2, 8, 0}, {2, 8, 1}, {1, 8, 1}, {2, 8, 1}};
    static const int t10[8][3] = {{2, 10, 0}, {1, 10, 0}, {2, 5, 0}, {2, 10, 0}, {2, 10, 0}, {2, 10, 1}, {1, 10, 1}, {2, 10, 1}};
    static const int t22[8][3] = {{2, 11, 0}, {1, 11, 0}, {1, 22, 0}, {2, 11, 0}, {2, 11, 0}, {1, 22, 1}, {1, 22, 1}, {1, 22, 1}};
    static const int t27[8][3] = {{2, 9, 0}, {1, 9, 0}, {1, 27, 0}, {2, 9, 0}, {2, 9, 0}, {1, 27, 1}, {1, 27, 1}, {1, 27, 1}};
    const int (*t)[3] = t8;
    const int cls = ns_class(C);
    switch (cls) { case 10: t = t10; break; case 22: t = t22; break; case 27: t = t27; break; default: break; }
    if (nru < 0 || nru > 7) nru = 0;
    if (cls == 0 && nru > 4) nru = 0;                  // the pipelined loop needs a compile-time slot count
    if ((pair && t[nru][0] != 2) || ((epi == EPI_ROPE_KV || epi == EPI_SWIGLU) && t[nru][0] > 2)) nru = t[nru][2] ? 5 : 0;
    *NR = t[nru][0]; *U = t[nru][1];
    if (pipe) *pipe = t[nru][2];
}

template <int PRO, int EPI, int NS>
static hipError_t launch_gemv_ns(int NR, int U, int pipe, const GemvArgs& a, int grid, bool nt, hipStream_t st) {
    constexpr bool pair = false;
    constexpr bool rope = EPI == EPI_ROPE_KV || EPI == EPI_SWIGLU;
#define THK_TRY(nr, u, pp)                                                                                  \
    if constexpr ((NS == 0 || NS % (u) == 0) && (!pair || (nr) == 2) && (!rope || (nr) <= 2) && (!(pp) || (NS != 0 && (u) == NS))) {   \
        if (NR == (nr) && U == (u) && pipe == (pp)) return launch_gemv_t<nr, u, NS, PRO, EPI, (pp) != 0>(a, grid, nt, st); \
    }
    if constexpr (NS == 8 || NS == 0) { THK_TRY(2, 8, 0) THK_TRY(1, 8, 0) THK_TRY(2, 4, 0) }
    if constexpr (NS == 8) { THK_TRY(2, 8, 1) THK_TRY(1, 8, 1) }
    if constexpr (NS == 10) { THK_TRY(2, 10, 0) THK_TRY(1, 10, 0) THK_TRY(2, 5, 0) THK_TRY(2, 10, 1) THK_TRY(1, 10, 1) }
    if constexpr (NS == 22) { THK_TRY(2, 11, 0) THK_TRY(1, 11, 0) THK_TRY(1, 22, 0) THK_TRY(1, 22, 1) }
    if constexpr (NS == 27) { THK_TRY(2, 9, 0) THK_TRY(1, 9, 0) THK_TRY(1, 27, 0) THK_TRY(1, 27, 1) }
#undef THK_TRY
    return hipErrorInvalidValue;
}

template <int PRO, int EPI>
static hipError_t launch_gemv_pe(int nru, const GemvArgs& a, int grid, bool nt, hipStream_t st) {
    int NR, U, pipe; gemv_variant(a.C, EPI, nru, &NR, &U, &pipe);
    switch (ns_class(a.C)) {
        case 8: return launch_gemv_ns<PRO, EPI, 8>(NR, U, pipe, a, grid, nt, st);
        case 10: return launch_gemv_ns<PRO, EPI, 10>(NR, U, pipe, a, grid, nt, st);
        case 22: return launch_gemv_ns<PRO, EPI, 22>(NR, U, pipe, a, grid, nt, st);
        case 27: return launch_gemv_ns<PRO, EPI, 27>(NR, U, pipe, a, grid, nt, st);
        default: return launch_gemv_ns<PRO, EPI, 0>(NR, U, pipe, a, grid, nt, st);
    }
}

// quarter-row form of y = resid + W x (gemv_quarter_body; tunable gemv_variant_w2 = 8; two rows in flight per wave - 9 - measured slower, retired)
template <int NS, int NR>
__global__ __launch_bounds__(kWaves * 64) void gemv_quarter_kernel(const uint16_t* W0, const float* x, const float* resid, float* y, int C, int R, int nblk, const GemvArgs a) {
    GemvArgs b = a;
    b.W[0] = W0; b.x = x; b.resid = resid; b.y = y; b.C = C; b.R = R;
    gemv_quarter_body<NS, NR>(b, blockIdx.x, nblk);
}
template <int NS, int NR>
static hipError_t launch_gemv_quarter_k(const GemvArgs& a, int grid, hipStream_t st) {
    const size_t smem = (size_t)NS * 512 * 4 + 128 + 512;
    auto kn = gemv_quarter_kernel<NS, NR>;
    static size_t attr_set[kMaxDevices] = {};
    if (smem > 48 * 1024) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
        if (smem > attr_set[dev]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return e;
            attr_set[dev] = smem;
        }
    }
    hipLaunchKernelGGL(kn, dim3(grid), dim3(kWaves * 64), smem, st, a.W[0], a.x, a.resid, a.y, a.C, a.R, grid, a);
    return hipGetLastError();
}
// rows per workgroup <= 32 (the LDS slots of the quarter sums); false = the shape / geometry has no quarter-row form
bool gemv_quarter_ok(int C, int R, int grid) {
    return (C == 4096 || C == 5120 || C == 11008 || C == 13824) && grid > 0 && (long)grid * 32 >= R;
}
hipError_t launch_gemv_quarter(const GemvArgs& a, int grid, hipStream_t st) {
    if (!gemv_quarter_ok(a.C, a.R, grid) || !a.resid) return hipErrorInvalidValue;
    switch (a.C) {
        case 4096: return launch_gemv_quarter_k<8, 1>(a, grid, st);
        case 5120: return launch_gemv_quarter_k<10, 1>(a, grid, st);
        case 11008: return launch_gemv_quarter_k<22, 1>(a, grid, st);
        default: return launch_gemv_quarter_k<27, 1>(a, grid, st);
    }
}

int gemv_rows_per_group(int C, int epi, int nru) {
    int NR, U; gemv_variant(C, epi, nru, &NR, &U, nullptr);
    return NR;
}

hipError_t launch_gemv(int pro, int epi, int nru, const GemvArgs& a, int grid, bool nt, hipStream_t st) {
    if (a.C < 256 || a.C % 256 != 0) return hipErrorInvalidValue;
    if (epi == EPI_ROPE_KV && pro == PRO_RMS) return launch_gemv_pe<PRO_RMS, EPI_ROPE_KV>(nru, a, grid, nt, st);
    if (epi == EPI_ROPE_KV && pro == PRO_RMS_EMBED) return (a.embed && a.tok_ptr && a.x_out) ? launch_gemv_pe<PRO_RMS_EMBED, EPI_ROPE_KV>(nru, a, grid, nt, st) : hipErrorInvalidValue;
    if (epi == EPI_SWIGLU && pro == PRO_RMS) return launch_gemv_pe<PRO_RMS, EPI_SWIGLU>(nru, a, grid, nt, st);
    if (epi == EPI_HEAD && pro == PRO_RMS) return launch_gemv_pe<PRO_RMS, EPI_HEAD>(nru, a, grid, nt, st);
    if (epi == EPI_HEAD && pro == PRO_COPY) return launch_gemv_pe<PRO_COPY, EPI_HEAD>(nru, a, grid, nt, st);
    if (epi == EPI_RESID && pro == PRO_ATTN) return launch_gemv_pe<PRO_ATTN, EPI_RESID>(nru, a, grid, nt, st);
    if (epi == EPI_RESID && pro == PRO_COPY) return launch_gemv_pe<PRO_COPY, EPI_RESID>(nru, a, grid, nt, st);
    if (epi == EPI_STORE && pro == PRO_COPY) return launch_gemv_pe<PRO_COPY, EPI_STORE>(nru, a, grid, nt, st);
    return hipErrorInvalidValue;
}


static_assert(KernargLead<decltype(&gemv_kernel<1, 8, 8, 1, 3, true, 0, true, 4>)>::bytes() == kKernargPreloadBytes &&
              KernargLead<decltype(&gemv_kernel<1, 8, 8, 0, 2, true, 0, true, 4>)>::bytes() == kKernargPreloadBytes,
              "gemv_kernel: the explicit scalars ahead of GemvArgs must fill exactly the 14 preloaded dwords");

// leading scalars: preloaded into SGPRs at wave launch (see gemv_kernel) - the position, q and the cache rows are what the chain
// of dependent loads starts from
template <int D, int WAVES, bool KVH, bool PIPE = false>
__global__ __launch_bounds__(WAVES * 64) void attn_decode_kernel(const int32_t* pos_ptr, const float* q, const float* kcache, const float* vcache,
                                                                 int pos_val, int H, int nsplit, int tc_signed, unsigned ns_magic, int nq, const AttnArgs a) {
    AttnArgs b = a;
    b.pos_ptr = pos_ptr; b.q = q; b.kcache = kcache; b.vcache = vcache; b.pos_val = pos_val; b.H = H; b.nsplit = nsplit; b.nq = nq;
    b.tc = tc_signed < 0 ? -tc_signed : tc_signed; b.tc_dyn = tc_signed < 0; b.ns_magic = ns_magic;     // (tc_dyn travels as the sign of tc: its slot carries the magic number)
    attn_body<D, WAVES, KVH, PIPE>(b, blockIdx.x);
}

template <int D, int WAVES>
static void launch_attn_dw(const AttnArgs& a, int grid, hipStream_t st) {
    const unsigned ns_magic = (unsigned)(((1ull << 32) + (unsigned)a.nsplit - 1) / (unsigned)a.nsplit);     // nsplit == 1: 0 (2^32 truncated) - attn_body does not divide then
#define THK_ATTN_GO(kvh, pp) hipLaunchKernelGGL((attn_decode_kernel<D, WAVES, kvh, pp>), dim3(grid), dim3(WAVES * 64), 0, st, a.pos_ptr, a.q, a.kcache, a.vcache, \
                                                    a.pos_val, a.H, a.nsplit, a.tc_dyn ? -a.tc : a.tc, ns_magic, a.nq, a)
    if constexpr (D == 128 && WAVES == 8) {       // the software-pipelined rounds (long caches) exist for the LLaMA head size, one workgroup per (head, split)
        if (a.pipe) { if (a.kv_f16) THK_ATTN_GO(true, true); else THK_ATTN_GO(false, true); return; }
    }
    if (a.kv_f16) THK_ATTN_GO(true, false); else THK_ATTN_GO(false, false);
#undef THK_ATTN_GO
}
int attn_round_positions(int D, int waves, bool kv_f16) {
    switch (D) {
        case 64: return waves * (kv_f16 ? AttnGeo<64, true>::PPW * AttnGeo<64, true>::UB : AttnGeo<64, false>::PPW * AttnGeo<64, false>::UB);
        case 256: return waves * (kv_f16 ? AttnGeo<256, true>::PPW * AttnGeo<256, true>::UB : AttnGeo<256, false>::PPW * AttnGeo<256, false>::UB);
        default: return waves * (kv_f16 ? AttnGeo<128, true>::PPW * AttnGeo<128, true>::UB : AttnGeo<128, false>::PPW * AttnGeo<128, false>::UB);
    }
}
hipError_t launch_attn_decode(const AttnArgs& a, hipStream_t st) {
    if (a.nsplit < 1 || a.nsplit > 4096 || a.tc <= 0) return hipErrorInvalidValue;
    const int grid = a.H * a.nsplit * (a.nq > 1 ? a.nq : 1);
    const bool w8 = a.waves == 8;
    switch (a.D) {
        case 64: if (w8) launch_attn_dw<64, 8>(a, grid, st); else launch_attn_dw<64, 4>(a, grid, st); break;
        case 128: if (a.waves == 16) launch_attn_dw<128, 16>(a, grid, st); else if (w8) launch_attn_dw<128, 8>(a, grid, st); else launch_attn_dw<128, 4>(a, grid, st); break;
        case 256: if (w8) launch_attn_dw<256, 8>(a, grid, st); else launch_attn_dw<256, 4>(a, grid, st); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Stand-alone combine of split partials -> out[H*D] (used by thk_attn_decode when nsplit > 1).
static_assert(KernargLead<decltype(&attn_decode_kernel<128, 8, false, false>)>::bytes() == kKernargPreloadBytes,
              "attn_decode_kernel: the explicit scalars ahead of AttnArgs must fill exactly the 14 preloaded dwords");
__global__ void attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml, float* out,
                                    int H, int D, int nsplit) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= H * D) return;
    const int h = e / D, d = e - h * D;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part_ml[(h * nsplit + s) * 2]);
    float L = 0.f, o = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float ms = part_ml[(h * nsplit + s) * 2];
        const float f = (ms == -INFINITY) ? 0.f : expf(ms - M);
        L += part_ml[(h * nsplit + s) * 2 + 1] * f; o += part_o[(size_t)(h * nsplit + s) * D + d] * f;
    }
    out[e] = o / L;
}
hipError_t launch_attn_combine(const float* part_o, const float* part_ml, float* out, int H, int D, int nsplit, hipStream_t st) {
    const int n = H * D;
    hipLaunchKernelGGL(attn_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, st, part_o, part_ml, out, H, D, nsplit);
    return hipGetLastError();
}

// ---------------------------------------------------------------- small element-wise kernels
// (stand-alone forms of K4,K5,K6,K10,K11,K12,K13 for the operator API; the model path uses the fused forms)
__global__ __launch_bounds__(kBlock) void rms_norm_kernel(float* x, int N) {
    __shared__ float red[4];
    float* row = x + (size_t)blockIdx.x * N;
    float ss = 0.f;
    for (int i = threadIdx.x; i < N; i += kBlock) ss += row[i] * row[i];
    ss = block_sum(ss, red);
    const float inv = 1.0f / sqrtf(ss / (float)N + 1e-6f);
    for (int i = threadIdx.x; i < N; i += kBlock) row[i] = row[i] * inv;
}
hipError_t launch_rms_norm(float* x, int rows, int N, hipStream_t st) {
    hipLaunchKernelGGL(rms_norm_kernel, dim3(rows), dim3(kBlock), 0, st, x, N);
    return hipGetLastError();
}

__global__ void row_mul_kernel(float* x, const float* __restrict__ g, int N, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) x[i] = x[i] * g[i % N];
}
hipError_t launch_row_mul(float* x, const float* g, int rows, int N, hipStream_t st) {
    const size_t total = (size_t)rows * N;
    hipLaunchKernelGGL(row_mul_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, g, N, total);
    return hipGetLastError();
}

// x viewed [n_tok, H, D]; one thread per (token, head, pair)
__global__ void rope_kernel(float* x, const float* __restrict__ tab, int n_tok, int H, int D, int n_past) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = D >> 1;
    if (i >= n_tok * H * half) return;
    const int jp = i % half, th = i / half, t = th / H;
    const float cs = tab[((size_t)(n_past + t) * half + jp) * 2], sn = tab[((size_t)(n_past + t) * half + jp) * 2 + 1];
    float* p = x + (size_t)th * D + 2 * jp;
    const float x0 = p[0], x1 = p[1];
    p[0] = x0 * cs - x1 * sn; p[1] = x0 * sn + x1 * cs;
}
hipError_t launch_rope(float* x, const float* tab, int n_tok, int H, int D, int n_past, hipStream_t st) {
    const int n = n_tok * H * (D / 2);
    hipLaunchKernelGGL(rope_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x, tab, n_tok, H, D, n_past);
    return hipGetLastError();
}

__global__ __launch_bounds__(kBlock) void row_softmax_kernel(float* x, int N) {
    __shared__ float red[4];
    float* row = x + (size_t)blockIdx.x * N;
    float mx = -1e14f;   // the reference's "-inf" (th.cpp:1867)
    for (int i = threadIdx.x; i < N; i += kBlock) mx = fmaxf(mx, row[i]);
    mx = wave_max(mx);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int i = threadIdx.x; i < N; i += kBlock) { const float e = expf(row[i] - mx); row[i] = e; sum += e; }
    sum = block_sum(sum, red);
    for (int i = threadIdx.x; i < N; i += kBlock) row[i] = row[i] / sum;
}
hipError_t launch_row_softmax(float* x, int rows, int N, hipStream_t st) {
    hipLaunchKernelGGL(row_softmax_kernel, dim3(rows), dim3(kBlock), 0, st, x, N);
    return hipGetLastError();
}

__global__ void add_kernel(const float* a, const float* b, float* c, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i] = a[i] + b[i];
}
__global__ void silu_kernel(float* a, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = a[i]; a[i] = v / (1.0f + expf(-v)); }
}
__global__ void mul_kernel(float* a, const float* b, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = a[i] * b[i];
}
hipError_t launch_add(const float* a, const float* b, float* c, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, c, n); return hipGetLastError();
}
hipError_t launch_silu(float* a, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(silu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, n); return hipGetLastError();
}
hipError_t launch_mul(float* a, const float* b, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, n); return hipGetLastError();
}

__global__ void kv_append_kernel(float* kc, float* vc, const float* k, const float* v, int pos, int E) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < E) { kc[(size_t)pos * E + i] = k[i]; vc[(size_t)pos * E + i] = v[i]; }
}
hipError_t launch_kv_append(float* kc, float* vc, const float* k, const float* v, int pos, int E, hipStream_t st) {
    hipLaunchKernelGGL(kv_append_kernel, dim3((E + 255) / 256), dim3(256), 0, st, kc, vc, k, v, pos, E); return hipGetLastError();
}

// ---------------------------------------------------------------- token plumbing
// Embedding fetch: x = f32(table[token,:]) (loader :185-195, th-llama.cpp:577-584).
__global__ __launch_bounds__(kBlock) void embed_kernel(const uint16_t* __restrict__ table, const SeqState* st, int token_val,
                                                       int E, float* x, unsigned long long* trace) {
    THK_STAMP(trace, 0, 0);
    const int token = st ? st->token : token_val;
    const _Float16* row = reinterpret_cast<const _Float16*>(table) + (size_t)token * E;
    for (int i = threadIdx.x; i < E; i += kBlock) x[i] = (float)row[i];
    THK_STAMP(trace, 0, 3);
}
__global__ __launch_bounds__(kBlock) void embed_rows_kernel(const uint16_t* __restrict__ table, const int32_t* __restrict__ tokens, int E, float* x) {
    const _Float16* row = reinterpret_cast<const _Float16*>(table) + (size_t)tokens[blockIdx.x] * E;
    for (int i = threadIdx.x; i < E; i += kBlock) x[(size_t)blockIdx.x * E + i] = (float)row[i];
}
hipError_t launch_embed_rows(const uint16_t* table, const int32_t* tokens_dev, int n, int E, float* x, hipStream_t st) {
    hipLaunchKernelGGL(embed_rows_kernel, dim3(n), dim3(kBlock), 0, st, table, tokens_dev, E, x);
    return hipGetLastError();
}
hipError_t launch_embed(const uint16_t* table, const SeqState* st_dev, int token_val, int E, float* x, hipStream_t st, unsigned long long* trace) {
    hipLaunchKernelGGL(embed_kernel, dim3(1), dim3(kBlock), 0, st, table, st_dev, token_val, E, x, trace);
    return hipGetLastError();
}

// Greedy pick + sequence bookkeeping after the head kernel: reduce the per-block
// best keys, write the token (first max wins, th-llama.cpp:826-838), log it,
// advance the position when asked.
// n_ctx > 0: the position only advances while pos + 1 < n_ctx, so a decode loop that outruns the host-side check
// (thk_model_decode_step(s) refuse it) can never index the caches or the RoPE table out of bounds.
// epoch != NULL: the engine's tag epoch is bumped here, i.e. after the engine launch of this step and before the next.
__global__ __launch_bounds__(kBlock) void finish_token_kernel(const FinishArgs a) { finish_token_body(a); }
hipError_t launch_finish_token(const unsigned long long* block_best, int nblocks, SeqState* st_dev, int32_t* gen_log, int log_cap,
                               const int* advance_ptr, int32_t* id_out, int n_ctx, unsigned* epoch, hipStream_t st, unsigned long long* trace,
                               unsigned long long* clock_log) {
    FinishArgs a{};
    a.block_best = block_best; a.nblocks = nblocks; a.st = st_dev; a.gen_log = gen_log; a.log_cap = log_cap; a.advance_ptr = advance_ptr;
    a.id_out = id_out; a.n_ctx = n_ctx; a.epoch = epoch; a.trace = trace; a.clock_log = clock_log;
    return launch_finish_token_args(a, st);
}
hipError_t launch_finish_token_args(const FinishArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(finish_token_kernel, dim3(1), dim3(kBlock), 0, st, a);
    return hipGetLastError();
}
// Non-head stages only advance the position (same clamp, same epoch bump).
__global__ void advance_pos_kernel(SeqState* st, const int* advance_ptr, int n_ctx, unsigned* epoch) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (*advance_ptr && (n_ctx <= 0 || st->pos + 1 < n_ctx)) st->pos += 1;
        if (epoch) *epoch += 1u;
    }
}
hipError_t launch_advance_pos(SeqState* st_dev, const int* advance_ptr, int n_ctx, unsigned* epoch, hipStream_t st) {
    hipLaunchKernelGGL(advance_pos_kernel, dim3(1), dim3(64), 0, st, st_dev, advance_ptr, n_ctx, epoch);
    return hipGetLastError();
}

// Arg-max over a logits vector already in memory (thk_argmax operator).
__global__ __launch_bounds__(kBlock) void argmax_kernel(const float* __restrict__ logits, int V, unsigned long long* block_best) {
    __shared__ unsigned long long sm[kBlock];
    unsigned long long b = 0ull;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < V; i += gridDim.x * kBlock) {
        const unsigned long long k = argmax_key(logits[i], (unsigned)i); b = k > b ? k : b;
    }
    sm[threadIdx.x] = b;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) { const unsigned long long o = sm[threadIdx.x + s]; if (o > sm[threadIdx.x]) sm[threadIdx.x] = o; }
        __syncthreads();
    }
    if (threadIdx.x == 0) block_best[blockIdx.x] = sm[0];
}
hipError_t launch_argmax(const float* logits, int V, unsigned long long* block_best, int nblocks, hipStream_t st) {
    hipLaunchKernelGGL(argmax_kernel, dim3(nblocks), dim3(kBlock), 0, st, logits, V, block_best);
    return hipGetLastError();
}

// ---------------------------------------------------------------- top-k of a logits vector (stochastic sampler, th-llama.cpp:814-907)
// keys[j] = (order-preserving map of logits[i]) << 32 | ~i  for the k largest, sorted descending: value descending, ties by
// ascending index - a total order, so the selection and its order are unique.  V <= 32768, k <= 1024.
constexpr int kTopkThreads = 1024, kTopkPer = 32, kTopkMax = 1024, kTopkChunk = 16384;     // a merge workgroup sorts <= 16384 keys (128 KiB of LDS)
// Round 5, two launches of one bitonic network.  (1) topk_local_kernel: every workgroup sorts 1024 keys in LDS (descending) and keeps its min(k, 1024)
// largest - the global top k is a subset of the union of the local ones - so 32 compute units share the work and the second launch sees 32 x k candidates;
// (2) topk_merge_kernel, ONE workgroup: sorts the candidates (<= 2048 for k <= 64; larger k: up to 32768 keys, several elements per thread) and writes the
// first k.  History of this kernel, each version slower than the 128 KB logits read-back it was meant to replace (46 us per token, tools/step_paths_probe.py):
// an 8-pass radix select whose first passes sent all 32000 LDS atomics to the two or three bins the logits' exponents share; a bisection on the key (64
// barrier steps, 65 us); a 16-ary search with all 32000 keys in ONE workgroup (114 us), then on 5000 pre-selected candidates (52-80 us) - a workgroup is one
// compute unit, 64 lanes per clock: 1024 threads x 600 instructions x 16 steps is 10 k cycles per step whatever the algorithm's elegance.
constexpr int kTopkLocal = 1024;
// descending bitonic sort of N keys (a power of two) in LDS by THREADS threads; ends with a barrier
template <int THREADS>
__device__ __forceinline__ void bitonic_desc(unsigned long long* sk, int N, int tid) {
    for (int size = 2; size <= N; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (N >> 1); t += THREADS) {             // compare-exchange number t of this step: elements i < p = i + stride
                const int i = ((t & ~(stride - 1)) << 1) | (t & (stride - 1)), p = i + stride;     // stride is a power of two: no division
                const bool desc = (i & size) == 0;
                const unsigned long long x = sk[i], y = sk[p];
                if ((x < y) == desc) { sk[i] = y; sk[p] = x; }
            }
            __syncthreads();
        }
}
__global__ __launch_bounds__(256) void topk_local_kernel(const float* __restrict__ logits, int V, int kk /* min(k, 1024) */, unsigned long long* __restrict__ cand) {
    __shared__ unsigned long long sk[kTopkLocal];
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < kTopkLocal / 256; ++j) {
        const int i = blockIdx.x * kTopkLocal + j * 256 + tid;
        sk[j * 256 + tid] = i < V ? argmax_key(logits[i], (unsigned)i) : 0ull;     // 0 is below every real key
    }
    __syncthreads();
    bitonic_desc<256>(sk, kTopkLocal, tid);
    for (int i = tid; i < kk; i += 256) cand[(size_t)blockIdx.x * kk + i] = sk[i];
}
// workgroup g sorts candidates [g * chunk, min(n_in, (g + 1) * chunk)) and writes its first k to out + g * k (one workgroup: the final keys, then `done`)
__global__ __launch_bounds__(kTopkThreads) void topk_merge_kernel(const unsigned long long* __restrict__ cand, int n_in, int chunk, int n_pow2 /* >= chunk */, int k,
                                                                  unsigned long long* __restrict__ out, unsigned long long* done /* host-mapped word or NULL */,
                                                                  unsigned long long epoch) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sm[];           // n_pow2 keys
    const int tid = threadIdx.x, i0 = blockIdx.x * chunk, n = min(chunk, n_in - i0);
    for (int i = tid; i < n_pow2; i += kTopkThreads) sm[i] = i < n ? cand[i0 + i] : 0ull;
    __syncthreads();
    bitonic_desc<kTopkThreads>(sm, n_pow2, tid);
    for (int i = tid; i < k; i += kTopkThreads) out[(size_t)blockIdx.x * k + i] = sm[i];
    if (done) {                                   // out is host-mapped: tell the polling host thread (every storing thread fences, then one publishes)
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// candidates of the local sorts + the intermediate level a large k needs (k > 512 at V = 32000: two merge levels)
size_t topk_scratch_bytes(int V, int k) {
    const size_t n_in = (size_t)((V + kTopkLocal - 1) / kTopkLocal) * (k < kTopkLocal ? k : kTopkLocal);
    return (n_in + ((n_in + kTopkChunk - 1) / kTopkChunk) * (size_t)k) * 8;
}
hipError_t launch_topk(const float* logits, int V, int k, unsigned long long* keys_out, unsigned long long* cand /* topk_scratch_bytes(V, k) */, hipStream_t st,
                       unsigned long long* done, unsigned long long epoch) {
    if (V < 1 || V > kTopkThreads * kTopkPer || k < 1 || k > kTopkMax || k > V || !cand) return hipErrorInvalidValue;
    const int kk = k < kTopkLocal ? k : kTopkLocal, nwg = (V + kTopkLocal - 1) / kTopkLocal;
    int n_in = nwg * kk;
    hipLaunchKernelGGL(topk_local_kernel, dim3(nwg), dim3(256), 0, st, logits, V, kk, cand);
    static bool attr_done[kMaxDevices] = {};
    int dev = 0; (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < kMaxDevices && !attr_done[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void*)topk_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kTopkChunk * 8);
        if (e != hipSuccess) return e;
        attr_done[dev] = true;
    }
    unsigned long long* src = cand;
    unsigned long long* mid = cand + n_in;                                // one intermediate level at most: ceil(32768 / 16384) * 1024 = 2048 candidates
    for (;;) {
        const int groups = (n_in + kTopkChunk - 1) / kTopkChunk, chunk = groups == 1 ? n_in : kTopkChunk;
        int n2 = 64;
        while (n2 < chunk) n2 <<= 1;
        const bool last = groups == 1;
        hipLaunchKernelGGL(topk_merge_kernel, dim3(groups), dim3(kTopkThreads), (size_t)n2 * 8, st, src, n_in, chunk, n2, k, last ? keys_out : mid, last ? done : nullptr, epoch);
        if (last) break;
        src = mid; n_in = groups * k;
    }
    return hipGetLastError();
}

bool trace_compiled() {
#ifdef THK_TRACE
    return true;
#else
    return false;
#endif
}

// ---------------------------------------------------------------- synthetic tensors
// Bit-identical twin of oracle/thk_oracle.c synth_value(): integer hash, one f32
// multiply (+ one add for gains), RNE f16 conversion.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// The product must be rounded to f32 BEFORE the f16 conversion / the +1 (two roundings, as on the
// CPU).  hipcc would otherwise contract mul+add into an fma and mul+cvt into a mixed-precision op
// (observed: 13 of 262,144 f16 values and 0.9 % of gains off by one ulp), so contraction is
// switched off here and the product is pinned in a VGPR.
__device__ __forceinline__ float synth_value(uint64_t key, uint64_t i, float scale) {
#pragma clang fp contract(off)
    const uint64_t h = splitmix64(key + i);
    const int s = (int)((h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48));
    float v = (float)(s - 131070) * scale;
    asm volatile("" : "+v"(v));
    return v;
}
__global__ void synth_f16_kernel(uint64_t key, float scale, size_t n, _Float16* out) {
#pragma clang fp contract(off)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (_Float16)synth_value(key, i, scale);
}
__global__ void synth_gain_kernel(uint64_t key, float scale, size_t n, float* out) {
#pragma clang fp contract(off)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float g = 1.0f + synth_value(key, i, scale);
        asm volatile("" : "+v"(g));
        out[i] = g;
    }
}
hipError_t launch_synth_f16(uint64_t key, float scale, size_t n, void* out, hipStream_t st) {
    const unsigned grid = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(synth_f16_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, key, scale, n, reinterpret_cast<_Float16*>(out));
    return hipGetLastError();
}
hipError_t launch_synth_gain(uint64_t key, float scale, size_t n, float* out, hipStream_t st) {
    const unsigned grid = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(synth_gain_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, key, scale, n, out);
    return hipGetLastError();
}

}  // namespace thk
