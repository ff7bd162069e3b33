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
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "thk_kernels.hpp"
#include "thk_device.hpp"

namespace thk {

// Development timeline (libthk_trace.so only, built with -DTHK_TRACE; tools/step_trace.py): every wave stamps the 100 MHz
// s_memrealtime counter at up to four points of its kernel into [workgroup][wave (8)][4].  Scalar instructions only (the stamp
// is written with s_store_dwordx2, flushed by s_dcache_wb at the last one): no VGPR, no exec-mask branch, so the register
// allocation and occupancy of the traced build stay those of the product build.  In the product build the macro is empty.
#ifdef THK_TRACE
__device__ __forceinline__ void thk_stamp(unsigned long long* tr, int bid, int slot) {
    if (tr) {                                                            // kernel argument: a scalar branch
        unsigned long long t;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
        const unsigned off = __builtin_amdgcn_readfirstlane(((unsigned)bid * 8u + (threadIdx.x >> 6)) * 32u + (unsigned)slot * 8u);
        asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(t), "s"(tr), "s"(off));
        if (slot == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
    }
}
// THK_TRACE=1 stamps kernel entry and exit only (register allocation identical to the product build, checked with
// -Rpass-analysis=kernel-resource-usage); THK_TRACE=2 adds the two inner stamps, which cost the mat-vec kernels 20+ VGPRs.
#define THK_STAMP(tr, bid, slot) do { if (THK_TRACE >= 2 || (slot) == 0 || (slot) == 3) thk_stamp((tr), (bid), (slot)); } while (0)
#else
#define THK_STAMP(tr, bid, slot) do { } while (0)
#endif

template <int WPB = kWaves>
__device__ __forceinline__ float block_sum(float v, float* red /* >= WPB floats of LDS */) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    if (WPB == 4) t = (red[0] + red[1]) + (red[2] + red[3]);
    else { t = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7])); }
    return t;
}

// ---------------------------------------------------------------- LDS x-vector layout
// A row is walked in "slots" of 64 lanes x 8 elements (1 KiB of f16 per wave-instruction).
// Lane l of slot c needs elements [(c*64+l)*8, +8).  They are stored as two float4
// arrays so consecutive lanes hit consecutive 16-byte slots (conflict-free ds_read_b128):
//   lo[c*64+l] = elems 0..3, hi[c*64+l] = elems 4..7 ; hi starts at ns*256 floats.
// The arrays are zero-padded to ns whole slots, so a lane past the end of a row (C % 512
// == 256) multiplies a clamped, valid weight vector by zeros: no divergent branch is
// needed in the streaming loop.
__device__ __forceinline__ int xs_index(int e, int ns) {   // float index in LDS for element e
    const int g = e >> 3, j = e & 7;
    return (j < 4 ? 0 : (ns << 8)) + (g << 2) + (j & 3);
}

// Store index for float4 #i of the padded vector; threads past the end (only possible when the
// slot count is odd) are steered to a dummy 16-byte slot behind the vector instead of branching.
__device__ __forceinline__ int xs_store_index(int i, int ns) {
    return (i < (ns << 7)) ? xs_index(i << 2, ns) : (ns << 9) + 8;
}

// ---------------------------------------------------------------- GEMV prologues
// All run with the whole block; on return xs[] holds the activation vector and a
// __syncthreads() has been executed.

// Each thread owns float4 #(tid + k*256), k < KP, of the (zero padded) vector.  Prologues are
// split in two phases so the kernel can order its memory traffic:
//   issue()  - fire all global loads of the activation vector (L2-resident, back to back)
//   [the kernel then fires the wave's first batch of WEIGHT loads]
//   finish() - wait for the activation loads only (vmcnt counts in order, so they had to be
//              issued first), reduce / combine, write LDS, __syncthreads()
// => the prologue's latency and math hide under the HBM latency of the first weight batch.
// With a run-time slot count (NS == 0) issue() is empty and finish() loops.
template <int NS, int WPB> struct PrologueK { static constexpr int value = NS ? (NS * 128 + WPB * 64 - 1) / (WPB * 64) : 1; };

// plain copy (th.cpp K1 with no fused producer)
template <int NS, int WPB>
struct ProCopy {
    static constexpr int KP = PrologueK<NS, WPB>::value;
    static constexpr int BT = WPB * 64;
    f4 v[KP];
    __device__ __forceinline__ void issue(const GemvArgs& a) {
        if (NS == 0) return;
#pragma unroll
        for (int k = 0; k < KP; ++k) v[k] = *reinterpret_cast<const f4*>(a.x + min((int)(threadIdx.x + k * BT) << 2, a.C - 4));
    }
    __device__ __forceinline__ void finish(const GemvArgs& a, float* xs, float*, int ns, int) {
        const int C = a.C;
        if (NS != 0) {
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int i = threadIdx.x + k * BT;
                const f4 o = ((i << 2) < C) ? v[k] : f4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f4*>(xs + xs_store_index(i, ns)) = o;
            }
        } else {
            for (int i = threadIdx.x; i < (ns << 7); i += BT) {
                f4 o = {0.f, 0.f, 0.f, 0.f};
                if ((i << 2) < C) o = *reinterpret_cast<const f4*>(a.x + (i << 2));
                *reinterpret_cast<f4*>(xs + xs_index(i << 2, ns)) = o;
            }
        }
        __syncthreads();
    }
};

// RMSNorm + gain (K4 th.cpp:1169-1198, K5 :1311-1313): xs = (x * inv) * g
// EMB: the input vector is the embedding row of the sequence's current token (loader :185-195, th-llama.cpp:577-584: x =
// f32(table[token,:])), fetched here instead of by a launch of its own; block 0 also writes the f32 row to a.x_out, which
// the layer's residual add reads two launches later.
template <int NS, bool EMB, int WPB>
struct ProRms {
    static constexpr int KP = PrologueK<NS, WPB>::value;
    static constexpr int BT = WPB * 64;
    f4 v[KP], g[KP];
    __device__ __forceinline__ f4 ldx(const GemvArgs& a, const _Float16* row, int ic) {
        if (EMB) {
            const h4 h = *reinterpret_cast<const h4*>(row + ic);
            return f4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        }
        return *reinterpret_cast<const f4*>(a.x + ic);
    }
    __device__ __forceinline__ void issue(const GemvArgs& a) {
        if (NS == 0) return;
        const _Float16* row = EMB ? reinterpret_cast<const _Float16*>(a.embed) + (size_t)(*a.tok_ptr) * a.C : nullptr;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int ic = min((int)(threadIdx.x + k * BT) << 2, a.C - 4);     // branch-free: clamp, select later
            v[k] = ldx(a, row, ic);
            g[k] = *reinterpret_cast<const f4*>(a.gain + ic);
        }
    }
    __device__ __forceinline__ void finish(const GemvArgs& a, float* xs, float* red, int ns, int bid) {
        const int C = a.C;
        if (NS != 0) {
            float ss = 0.f;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int i = threadIdx.x + k * BT;
                if ((i << 2) >= C) v[k] = f4{0.f, 0.f, 0.f, 0.f};
                else if (EMB && bid == 0) *reinterpret_cast<f4*>(a.x_out + (i << 2)) = v[k];
                ss += v[k].x * v[k].x + v[k].y * v[k].y + v[k].z * v[k].z + v[k].w * v[k].w;
            }
            ss = block_sum<WPB>(ss, red);
            const float inv = 1.0f / sqrtf(ss / (float)C + 1e-6f);
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int i = threadIdx.x + k * BT;
                f4 o;
                o.x = (v[k].x * inv) * g[k].x; o.y = (v[k].y * inv) * g[k].y; o.z = (v[k].z * inv) * g[k].z; o.w = (v[k].w * inv) * g[k].w;
                *reinterpret_cast<f4*>(xs + xs_store_index(i, ns)) = o;
            }
        } else {
            const _Float16* row = EMB ? reinterpret_cast<const _Float16*>(a.embed) + (size_t)(*a.tok_ptr) * C : nullptr;
            float ss = 0.f;
            for (int i = threadIdx.x; i < (ns << 7); i += BT) {
                f4 t = {0.f, 0.f, 0.f, 0.f};
                if ((i << 2) < C) {
                    t = ldx(a, row, i << 2);
                    if (EMB && bid == 0) *reinterpret_cast<f4*>(a.x_out + (i << 2)) = t;
                }
                ss += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
                *reinterpret_cast<f4*>(xs + xs_index(i << 2, ns)) = t;   // raw copy, normalised below
            }
            ss = block_sum<WPB>(ss, red);
            const float inv = 1.0f / sqrtf(ss / (float)C + 1e-6f);
            for (int i = threadIdx.x; i < (C >> 2); i += BT) {
                const f4 gg = *reinterpret_cast<const f4*>(a.gain + (i << 2));
                f4* p = reinterpret_cast<f4*>(xs + xs_index(i << 2, ns));
                f4 t = *p;   // same thread wrote it
                t.x = (t.x * inv) * gg.x; t.y = (t.y * inv) * gg.y; t.z = (t.z * inv) * gg.z; t.w = (t.w * inv) * gg.w;
                *p = t;
            }
        }
        __syncthreads();
    }
};

// Attention split combine: xs[h*D+d] = sum_s o_s[d] * e^{m_s-M} / sum_s l_s * e^{m_s-M}
// NSP = compile-time split count so all 2*NSP loads of a float4 are issued together.
template <int NSP>
__device__ __forceinline__ f4 attn_merge(const float2 (&ml)[NSP], const f4 (&ov)[NSP]) {
    float M = -INFINITY;
#pragma unroll
    for (int s = 0; s < NSP; ++s) M = fmaxf(M, ml[s].x);
    float L = 0.f; f4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NSP; ++s) {
        const float sc = (ml[s].x == -INFINITY) ? 0.f : expf(ml[s].x - M);
        L += ml[s].y * sc; o += ov[s] * sc;
    }
    return o * (1.0f / L);
}
template <int NS, int NSP, int WPB>
struct ProAttn {
    static constexpr int KP = PrologueK<NS, WPB>::value;
    static constexpr int BT = WPB * 64;
    float2 ml[KP][NSP];
    f4 ov[KP][NSP];
    __device__ __forceinline__ void load1(const GemvArgs& a, int e, float2 (&m)[NSP], f4 (&o)[NSP]) {
        const int h = e / a.D, d = e - h * a.D;
#pragma unroll
        for (int s = 0; s < NSP; ++s) {
            const float* pm = a.part_ml + (h * NSP + s) * 2;
            const float* po = a.part_o + (size_t)(h * NSP + s) * a.D + d;
            m[s] = *reinterpret_cast<const float2*>(pm);
            o[s] = *reinterpret_cast<const f4*>(po);
        }
    }
    __device__ __forceinline__ void issue(const GemvArgs& a) {
        if (NS == 0) return;
#pragma unroll
        for (int k = 0; k < KP; ++k) load1(a, min((int)(threadIdx.x + k * BT) << 2, a.C - 4), ml[k], ov[k]);
    }
    __device__ __forceinline__ void finish(const GemvArgs& a, float* xs, float*, int ns, int) {
        const int C = a.C;
        if (NS != 0) {
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int i = threadIdx.x + k * BT;
                f4 res = attn_merge<NSP>(ml[k], ov[k]);
                if ((i << 2) >= C) res = f4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f4*>(xs + xs_store_index(i, ns)) = res;
            }
        } else {
            for (int i = threadIdx.x; i < (ns << 7); i += BT) {
                f4 res = {0.f, 0.f, 0.f, 0.f};
                if ((i << 2) < C) { float2 m[NSP]; f4 o[NSP]; load1(a, i << 2, m, o); res = attn_merge<NSP>(m, o); }
                *reinterpret_cast<f4*>(xs + xs_index(i << 2, ns)) = res;
            }
        }
        __syncthreads();
    }
};
template <int NS, int PRO, int NSP, int WPB> struct ProSelect { typedef ProCopy<NS, WPB> type; };
template <int NS, int NSP, int WPB> struct ProSelect<NS, GEMV_PRO_RMS, NSP, WPB> { typedef ProRms<NS, false, WPB> type; };
template <int NS, int NSP, int WPB> struct ProSelect<NS, GEMV_PRO_RMS_EMBED, NSP, WPB> { typedef ProRms<NS, true, WPB> type; };
template <int NS, int NSP, int WPB> struct ProSelect<NS, GEMV_PRO_ATTN, NSP, WPB> { typedef ProAttn<NS, (NSP > 0 ? NSP : 1), WPB> type; };

// ---------------------------------------------------------------- GEMV core
enum { PRO_COPY = GEMV_PRO_COPY, PRO_RMS = GEMV_PRO_RMS, PRO_ATTN = GEMV_PRO_ATTN, PRO_RMS_EMBED = GEMV_PRO_RMS_EMBED };
enum { EPI_STORE = GEMV_EPI_STORE, EPI_RESID = GEMV_EPI_RESID, EPI_ROPE_KV = GEMV_EPI_ROPE_KV, EPI_SWIGLU = GEMV_EPI_SWIGLU,
       EPI_HEAD = GEMV_EPI_HEAD };

template <bool NT>
__device__ __forceinline__ h8 ldw(const h8* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

// One launch = one fused op.  Work unit = "row group": NR weight rows streamed
// together by one wave.  Group g of EPI_* means:
//   STORE/RESID/HEAD : rows NR*g .. NR*g+NR-1 of W[0]
//   ROPE_KV (NR=2)   : rows 2g,2g+1 of the virtual [3E,E] stack W[0]=wq,W[1]=wk,W[2]=wv
//   SWIGLU (NR=2)    : row g of W[0]=w1 and row g of W[1]=w3
// NS = compile-time slot count (C = NS*512 or NS*512-256), 0 = run-time (any C % 256 == 0).
// U = slots per load batch (NS % U == 0 when NS != 0): NR*U 16-byte loads per lane are issued
// back to back with no intervening branch or wait.
// bid / nblk: this block's index and the number of blocks working on the op (== blockIdx.x / gridDim.x).
template <int NR, int U, int NS, int PRO, int EPI, bool NT, int NSP, bool PIPE, int WPB = kWaves>
__device__ __forceinline__ void gemv_body(const GemvArgs& a, const int bid, const int nblk) {
    static_assert(!PIPE || (NS != 0 && U == NS), "the pipelined loop keeps one whole row group in flight");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int C = a.C;
    const int nvec = C >> 3;                                  // 16-byte vectors per row
    const int ns = NS ? NS : ((nvec + 63) >> 6);
    float* xs = smem;                 // ns*512 floats
    float* red = smem + (ns << 9);    // floats 0-7: reduction, 8-11: dummy store slot, 12-27: EPI_HEAD scratch (one u64 per wave)
    // the wave index is read into an SGPR: row numbers and row pointers become scalar, so every weight load is
    // `global_load_dwordx4 v, v_lane_offset, s[row]` instead of carrying a 64-bit address per lane
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave_global = bid * WPB + wave;
    const int total_waves = nblk * WPB;
    const f4* xlo = reinterpret_cast<const f4*>(xs);
    const f4* xhi = reinterpret_cast<const f4*>(xs + (ns << 8));
    const int half_c = C >> 1;
    const int vlast = nvec - 1;

    // row pointers of group g (wave-uniform; independent of the activation vector)
    auto row_ptrs = [&](int g, const h8* (&rp)[NR]) {
        if (EPI == EPI_ROPE_KV) {
            const int r0 = 2 * g, which = r0 / a.E, rr = r0 - which * a.E;
            const uint16_t* base = which == 0 ? a.W[0] : (which == 1 ? a.W[1] : a.W[2]);
            rp[0] = reinterpret_cast<const h8*>(base + (size_t)rr * C);
            rp[1 % NR] = reinterpret_cast<const h8*>(base + (size_t)(rr + 1) * C);
        } else if (EPI == EPI_SWIGLU) {
            rp[0] = reinterpret_cast<const h8*>(a.W[0] + (size_t)g * C);
            rp[1 % NR] = reinterpret_cast<const h8*>(a.W[1] + (size_t)g * C);
        } else {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                int row = NR * g + r; if (row >= a.R) row = a.R - 1;   // tail rows recomputed, not stored
                rp[r] = reinterpret_cast<const h8*>(a.W[0] + (size_t)row * C);
            }
        }
    };
    // issue the NR*U 16-byte loads of slots [c0, c0+U) back to back (no branch, no wait)
    auto load_batch = [&](const h8* const (&rp)[NR], int c0, h8 (&w)[NR][U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = (c0 + u) * 64 + lane;
            const int vc = (NS == 0 || c0 + u == NS - 1) ? min(v, vlast) : v;   // only the last slot can overrun
#pragma unroll
            for (int r = 0; r < NR; ++r) w[r][u] = ldw<NT>(rp[r] + vc);
        }
    };
    auto compute_batch = [&](int c0, const h8 (&w)[NR][U], float (&acc)[NR], float (&acc_hi)[NR]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NS == 0 && c0 + u >= ns) break;                 // run-time slot count: wave-uniform
            const int v = (c0 + u) * 64 + lane;
            const f4 xl = xlo[v], xh = xhi[v];
            if (EPI == EPI_HEAD) {
                const bool hi = (v << 3) >= half_c;             // second K half (th.cpp:3549-3568)
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const float p = dot8(w[r][u], xl, xh, 0.f);
                    acc[r] += hi ? 0.f : p; acc_hi[r] += hi ? p : 0.f;
                }
            } else {
#pragma unroll
                for (int r = 0; r < NR; ++r) acc[r] = dot8(w[r][u], xl, xh, acc[r]);
            }
        }
    };

    unsigned long long best = 0ull;   // EPI_HEAD running arg-max of this wave (valid in lane 0)

    // Epilogue operands of a group (residuals / RoPE cos,sin).  The pipelined loop fetches them BEFORE it refills the ring with
    // the next group's weights: vmcnt retires in order, so a load issued behind the refill could only be waited for by draining
    // the whole prefetch.
    struct EpiOps { float resid[NR]; float cs, sn; };
    const int pos_pipe = (PIPE && EPI == EPI_ROPE_KV) ? (a.pos_ptr ? *a.pos_ptr : a.pos_val) : 0;
    auto epi_fetch = [&](int g, EpiOps& eo) {
        if (EPI == EPI_RESID) {
#pragma unroll
            for (int r = 0; r < NR; ++r) eo.resid[r] = a.resid[min(NR * g + r, a.R - 1)];      // wave-uniform address: one request
        } else if (EPI == EPI_ROPE_KV) {
            const int r0 = 2 * g, which = r0 / a.E, rr = r0 - which * a.E, j = rr % a.D;
            const float2 t = *reinterpret_cast<const float2*>(a.rope_tab + ((size_t)pos_pipe * (a.D >> 1) + (j >> 1)) * 2);
            eo.cs = t.x; eo.sn = t.y;
        }
    };
    auto finish_group = [&](int g, float (&acc)[NR], float (&acc_hi)[NR], const EpiOps* eo = nullptr) {
#pragma unroll
        for (int r = 0; r < NR; ++r) { acc[r] = wave_sum(acc[r]); if (EPI == EPI_HEAD) acc_hi[r] = wave_sum(acc_hi[r]); }
        if (EPI == EPI_STORE) {
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < NR; ++r) if (NR * g + r < a.R) a.y[NR * g + r] = acc[r];
            }
        } else if (EPI == EPI_RESID) {       // K11 th.cpp:2136-2147: c = a + b
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < NR; ++r) if (NR * g + r < a.R) a.y[NR * g + r] = (eo ? eo->resid[r] : a.resid[NR * g + r]) + acc[r];
            }
        } else if (EPI == EPI_ROPE_KV) {     // K6 th.cpp:1476-1490 + K/V append th-llama.cpp:332-339
            if (lane == 0) {
                const int pos = PIPE ? pos_pipe : (a.pos_ptr ? *a.pos_ptr : a.pos_val);
                const int r0 = 2 * g, which = r0 / a.E, rr = r0 - which * a.E;
                float y0 = acc[0], y1 = acc[1 % NR];
                if (which < 2) {
                    const int j = rr % a.D;    // even
                    const float cs = eo ? eo->cs : a.rope_tab[((size_t)pos * (a.D >> 1) + (j >> 1)) * 2];
                    const float sn = eo ? eo->sn : a.rope_tab[((size_t)pos * (a.D >> 1) + (j >> 1)) * 2 + 1];
                    const float t0 = y0 * cs - y1 * sn, t1 = y0 * sn + y1 * cs;
                    y0 = t0; y1 = t1;
                }
                if (which != 0 && a.kv_f16) {     // optional f16 cache: RNE rounding at the append (v_cvt_f16_f32)
                    _Float16* dh = reinterpret_cast<_Float16*>(which == 1 ? a.kcache : a.vcache) + (size_t)pos * a.E;
                    dh[rr] = (_Float16)y0; dh[rr + 1] = (_Float16)y1;
                } else {
                    float* dst = which == 0 ? a.y : (which == 1 ? a.kcache + (size_t)pos * a.E : a.vcache + (size_t)pos * a.E);
                    dst[rr] = y0; dst[rr + 1] = y1;
                }
            }
        } else if (EPI == EPI_SWIGLU) {      // K12 th.cpp:2706-2707, K13 :2512-2524
            if (lane == 0) { const float u1 = acc[0]; a.y[g] = (u1 / (1.0f + expf(-u1))) * acc[1 % NR]; }
        } else {                              // EPI_HEAD: K3 th.cpp:3926-3943 (+Q1 switch)
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int row = NR * g + r;
                    if (row < a.R) {
                        const bool covered = !a.lm_faithful || (row % a.q1_split) < a.q1_cov;
                        const float v = covered ? acc[r] + acc_hi[r] : acc[r];
                        a.y[row] = v;
                        const unsigned long long k = argmax_key(v, (unsigned)row);
                        best = k > best ? k : best;
                    }
                }
            }
        }
    };

    THK_STAMP(a.trace, bid, 0);
    // --- memory traffic is ordered: activation loads, then the wave's first weight batch (weights
    // do not depend on the activations), then the prologue math while the weights are in flight.
    int g = wave_global;
    const bool has_first = g < a.n_groups;
    const h8* rp0[NR];
    h8 w0[NR][U];
    row_ptrs(has_first ? g : a.n_groups - 1, rp0);   // idle waves (more waves than groups) load a valid row:
    typename ProSelect<NS, PRO, NSP, WPB>::type pro;      // an unconditional load keeps the vmcnt bookkeeping exact
    pro.issue(a);
    __builtin_amdgcn_sched_barrier(0);
    load_batch(rp0, 0, w0);
    __builtin_amdgcn_sched_barrier(0);
    pro.finish(a, xs, red, ns, bid);
    THK_STAMP(a.trace, bid, 1);

    if constexpr (PIPE) {
        // Software-pipelined stream: the ring w0 holds one whole row group (NR*NS 16-byte loads per lane).  Slot c of the NEXT
        // group is requested as soon as slot c of the current one has been consumed, so the wave keeps a constant NR*NS KiB in
        // flight from its first load to its last - no drain between batches or groups even at one wave per SIMD.
        auto slot_loads = [&](const h8* const (&rp)[NR], int c, h8 (&w)[NR][U]) {
            const int v = c * 64 + lane;
            const int vc = (c == NS - 1) ? min(v, vlast) : v;
#pragma unroll
            for (int r = 0; r < NR; ++r) w[r][c] = ldw<NT>(rp[r] + vc);
        };
        auto slot_fma = [&](int c, const h8 (&w)[NR][U], float (&acc)[NR], float (&acc_hi)[NR]) {
            const int v = c * 64 + lane;
            const f4 xl = xlo[v], xh = xhi[v];
            if (EPI == EPI_HEAD) {
                const bool hi = (v << 3) >= half_c;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const float p = dot8(w[r][c], xl, xh, 0.f);
                    acc[r] += hi ? 0.f : p; acc_hi[r] += hi ? p : 0.f;
                }
            } else {
#pragma unroll
                for (int r = 0; r < NR; ++r) acc[r] = dot8(w[r][c], xl, xh, acc[r]);
            }
        };
        if (has_first) {
            for (int gn = g + total_waves; gn < a.n_groups; g = gn, gn += total_waves) {     // steady state: every consume is followed by a refill
                const h8* rpn[NR];
                row_ptrs(gn, rpn);
                EpiOps eo;
                epi_fetch(g, eo);
                float acc[NR], acc_hi[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) { acc[r] = 0.f; acc_hi[r] = 0.f; }
                __builtin_amdgcn_sched_barrier(0);                                            // the operand loads stay ahead of the refills
#pragma unroll
                for (int c = 0; c < NS; ++c) {
                    slot_fma(c, w0, acc, acc_hi);
                    __builtin_amdgcn_sched_barrier(0);                                        // keep refill c right behind consume c
                    slot_loads(rpn, c, w0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                finish_group(g, acc, acc_hi, &eo);
            }
            EpiOps eo;                                                                        // last group: nothing left to request
            epi_fetch(g, eo);
            __builtin_amdgcn_sched_barrier(0);
            float acc[NR], acc_hi[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) { acc[r] = 0.f; acc_hi[r] = 0.f; }
#pragma unroll
            for (int c = 0; c < NS; ++c) slot_fma(c, w0, acc, acc_hi);
            THK_STAMP(a.trace, bid, 2);
            finish_group(g, acc, acc_hi, &eo);
        }
    } else {
    if (has_first) {
            float acc[NR], acc_hi[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) { acc[r] = 0.f; acc_hi[r] = 0.f; }
            compute_batch(0, w0, acc, acc_hi);
            THK_STAMP(a.trace, bid, 2);
            if (NS != 0) {
#pragma unroll
                for (int c0 = U; c0 < NS; c0 += U) {
                    h8 w[NR][U];
                    load_batch(rp0, c0, w);
                    __builtin_amdgcn_sched_barrier(0);   // keep all NR*U loads ahead of the first use
                    compute_batch(c0, w, acc, acc_hi);
                }
            } else {
                for (int c0 = U; c0 < ns; c0 += U) {
                    h8 w[NR][U];
                    load_batch(rp0, c0, w);
                    __builtin_amdgcn_sched_barrier(0);
                    compute_batch(c0, w, acc, acc_hi);
                }
            }
            finish_group(g, acc, acc_hi);
            g += total_waves;
        }
        for (; g < a.n_groups; g += total_waves) {
            const h8* rp[NR];
            row_ptrs(g, rp);
            float acc[NR], acc_hi[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) { acc[r] = 0.f; acc_hi[r] = 0.f; }
            if (NS != 0) {
#pragma unroll
                for (int c0 = 0; c0 < NS; c0 += U) {
                    h8 w[NR][U];
                    load_batch(rp, c0, w);
                    __builtin_amdgcn_sched_barrier(0);
                    compute_batch(c0, w, acc, acc_hi);
                }
            } else {
                for (int c0 = 0; c0 < ns; c0 += U) {
                    h8 w[NR][U];
                    load_batch(rp, c0, w);
                    __builtin_amdgcn_sched_barrier(0);
                    compute_batch(c0, w, acc, acc_hi);
                }
            }
            finish_group(g, acc, acc_hi);
        }
    }
    THK_STAMP(a.trace, bid, 3);
    if (EPI == EPI_HEAD) {
        unsigned long long* wb = reinterpret_cast<unsigned long long*>(red + 12);   // 16-byte aligned
        __syncthreads();
        if (lane == 0) wb[wave] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long b = wb[0];
            for (int w = 1; w < WPB; ++w) b = wb[w] > b ? wb[w] : b;
            a.block_best[bid] = b;
        }
    }
}

template <int NR, int U, int NS, int PRO, int EPI, bool NT, int NSP, bool PIPE, int WPB>
__global__ __launch_bounds__(WPB * 64) void gemv_kernel(const GemvArgs a) {
    gemv_body<NR, U, NS, PRO, EPI, NT, NSP, PIPE, WPB>(a, blockIdx.x, gridDim.x);
}

template <int NR, int U, int NS, int PRO, int EPI, int NSP, bool PIPE, int WPB>
static hipError_t launch_gemv_k(const GemvArgs& a, int grid, bool nt, hipStream_t st) {
    const int ns = NS ? NS : (((a.C >> 3) + 63) >> 6);
    const size_t smem = (size_t)ns * 512 * 4 + 128;
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
    hipLaunchKernelGGL(kn, dim3(grid), dim3(WPB * 64), smem, st, a);
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
//   0-3  batch loop: NR*U loads, then their FMAs, per batch
//   4    batch loop with four whole rows per wave in ONE batch (4096/5120 columns)
//   5-7  software-pipelined loop (gemv_body PIPE): one whole row group in flight, slot c of the next group requested as soon as
//        slot c of the current one is consumed.  5 = row pair (one row for 11008/13824 columns), 6 = one row, 7 = four rows
void gemv_variant(int C, int epi, int nru, int* NR, int* U, int* pipe) {
    const bool pair = (epi == EPI_ROPE_KV || epi == EPI_SWIGLU);
    static const int t8[8][3] = {{2, 8, 0}, {1, 8, 0}, {2, 4, 0}, {4, 4, 0}, {4, 8, 0}, {2, 8, 1}, {1, 8, 1}, {4, 8, 1}};
    static const int t10[8][3] = {{2, 10, 0}, {1, 10, 0}, {2, 5, 0}, {4, 5, 0}, {4, 10, 0}, {2, 10, 1}, {1, 10, 1}, {4, 10, 1}};
    static const int t22[8][3] = {{2, 11, 0}, {1, 11, 0}, {1, 22, 0}, {2, 11, 0}, {2, 11, 0}, {1, 22, 1}, {1, 22, 1}, {1, 22, 1}};
    static const int t27[8][3] = {{2, 9, 0}, {1, 9, 0}, {1, 27, 0}, {2, 9, 0}, {2, 9, 0}, {1, 27, 1}, {1, 27, 1}, {1, 27, 1}};
    const int (*t)[3] = t8;
    const int cls = ns_class(C);
    switch (cls) { case 10: t = t10; break; case 22: t = t22; break; case 27: t = t27; break; default: break; }
    if (nru < 0 || nru > 7) nru = 0;
    if (cls == 0 && nru > 4) nru = 0;                  // the pipelined loop needs a compile-time slot count
    if (cls == 0 && nru == 4) nru = 3;
    if (pair && t[nru][0] != 2) nru = t[nru][2] ? 5 : 0;
    *NR = t[nru][0]; *U = t[nru][1];
    if (pipe) *pipe = t[nru][2];
}

template <int PRO, int EPI, int NS>
static hipError_t launch_gemv_ns(int NR, int U, int pipe, const GemvArgs& a, int grid, bool nt, hipStream_t st) {
    constexpr bool pair = (EPI == EPI_ROPE_KV || EPI == EPI_SWIGLU);
#define THK_TRY(nr, u, pp)                                                                                  \
    if constexpr ((NS == 0 || NS % (u) == 0) && (!pair || (nr) == 2) && (!(pp) || (NS != 0 && (u) == NS))) {   \
        if (NR == (nr) && U == (u) && pipe == (pp)) return launch_gemv_t<nr, u, NS, PRO, EPI, (pp) != 0>(a, grid, nt, st); \
    }
    if constexpr (NS == 8 || NS == 0) { THK_TRY(2, 8, 0) THK_TRY(1, 8, 0) THK_TRY(2, 4, 0) THK_TRY(4, 4, 0) }
    if constexpr (NS == 8) { THK_TRY(4, 8, 0) THK_TRY(2, 8, 1) THK_TRY(1, 8, 1) THK_TRY(4, 8, 1) }
    if constexpr (NS == 10) { THK_TRY(2, 10, 0) THK_TRY(1, 10, 0) THK_TRY(2, 5, 0) THK_TRY(4, 5, 0) THK_TRY(4, 10, 0) THK_TRY(2, 10, 1) THK_TRY(1, 10, 1) THK_TRY(4, 10, 1) }
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

// ---------------------------------------------------------------- attention (decode)
// grid = H * nsplit blocks; block (h, s) owns positions [s*tc, (s+1)*tc) of head h.
// A position's head slice is D contiguous floats; D/4 lanes x float4 cover it, so a
// wave-instruction fetches PPW = 64/(D/4) positions.  Each wave runs an online
// softmax over its positions, the four waves are merged through LDS, and the block
// writes (m, l, o[D]) for the split (or the normalised output when nsplit == 1).
// Scores: S = (q.k) * 1/sqrt(D) scaled after the sum (th.cpp:527-529, th-llama.cpp:518);
// softmax K10 th.cpp:1901-1957.
// WAVES waves per block share one (head, split): more waves = fewer positions per wave, so every
// wave needs a single load batch (one HBM round trip) at T = 512 with 4 splits.
// one position's 4-element slice for this lane: f32 cache (16 bytes) or f16 cache (8 bytes, widened by v_cvt_f32_f16)
template <bool KVH>
__device__ __forceinline__ f4 ld_kv4(const float* base, size_t elem_off) {
    if (KVH) {
        const h4 h = __builtin_nontemporal_load(reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(base) + elem_off));
        return f4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    }
    return __builtin_nontemporal_load(reinterpret_cast<const f4*>(base + elem_off));
}
template <int D, int WAVES, bool KVH>
__device__ __forceinline__ void attn_body(const AttnArgs& a, const int bid) {
    constexpr int LPP = D / 4;          // lanes per position
    constexpr int PPW = 64 / LPP;       // positions per wave-instruction
    constexpr int UB = 8;               // wave-instructions per batch (K and V each)
    __shared__ float sm_o[WAVES][D];
    __shared__ float sm_ml[WAVES][2];

    // prefill: nq > 1 causal queries share one launch; query qi sits at position pos + qi
    const int per_q = a.H * a.nsplit;
    const int qi = a.nq > 1 ? bid / per_q : 0;
    const int hb = bid - qi * per_q;
    const int h = hb / a.nsplit, s = hb - h * a.nsplit;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / LPP, li = lane - grp * LPP;
    THK_STAMP(a.trace, bid, 0);
    const int T = (a.pos_ptr ? *a.pos_ptr : a.pos_val) + qi + 1;
    const int E = a.H * D;
    const int t0 = s * a.tc, t1 = min(t0 + a.tc, T);

    const f4 q = *reinterpret_cast<const f4*>(a.q + (size_t)qi * E + h * D + li * 4);
    const size_t hoff = (size_t)(h * D + li * 4);      // element offset of this lane's slice inside a cache row

    float m = -INFINITY, l = 0.f;
    f4 o = {0.f, 0.f, 0.f, 0.f};
    // wave w takes positions t0 + (it*WAVES + w)*PPW*UB + u*PPW + grp.  Loads are branch-free:
    // positions past the end are clamped to a valid row and masked out of the softmax.
    // (Round 3 tried fetching the first batch BEFORE the device-resident position is known - every row below n_ctx is
    // allocated - to take the position's round trip off the critical path: 0.6 us per launch SLOWER on MI355X, removed.)
    for (int tb = t0 + wave * (PPW * UB); tb < t1; tb += WAVES * PPW * UB) {
        f4 kv[UB], vv[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int t = min(tb + u * PPW + grp, t1 - 1);
            kv[u] = ld_kv4<KVH>(a.kcache, (size_t)t * E + hoff);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int t = min(tb + u * PPW + grp, t1 - 1);
            vv[u] = ld_kv4<KVH>(a.vcache, (size_t)t * E + hoff);
        }
        __builtin_amdgcn_sched_barrier(0);
        float sc[UB];
        float bm = -INFINITY;
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int t = tb + u * PPW + grp;
            float d = q.x * kv[u].x + q.y * kv[u].y + q.z * kv[u].z + q.w * kv[u].w;
            d = group_sum<LPP>(d) * a.scale;
            sc[u] = (t < t1) ? d : -INFINITY;
            bm = fmaxf(bm, sc[u]);
        }
        bm = wave_max(bm);                      // wave-uniform, finite (tb < t1 => lane group 0 valid)
        const float mn = fmaxf(m, bm);
        const float alpha = (m == -INFINITY) ? 0.f : expf(m - mn);
        l *= alpha; o *= alpha;
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const float p = expf(sc[u] - mn);   // exp(-inf) == 0 for masked positions
            l += p; o += vv[u] * p;
        }
        m = mn;
        THK_STAMP(a.trace, bid, 1);
    }
    // merge the PPW lane groups of the wave (same m): sum l and o across groups
    if (PPW >= 2) { l += __shfl_xor(l, LPP); o.x += __shfl_xor(o.x, LPP); o.y += __shfl_xor(o.y, LPP); o.z += __shfl_xor(o.z, LPP); o.w += __shfl_xor(o.w, LPP); }
    if (PPW >= 4) { l += __shfl_xor(l, 2 * LPP); o.x += __shfl_xor(o.x, 2 * LPP); o.y += __shfl_xor(o.y, 2 * LPP); o.z += __shfl_xor(o.z, 2 * LPP); o.w += __shfl_xor(o.w, 2 * LPP); }
    if (lane < LPP) *reinterpret_cast<f4*>(&sm_o[wave][lane * 4]) = o;
    if (lane == 0) { sm_ml[wave][0] = m; sm_ml[wave][1] = l; }
    __syncthreads();
    THK_STAMP(a.trace, bid, 2);
    if (threadIdx.x < D) {
        const int d = threadIdx.x;
        float M = -INFINITY;
        for (int w = 0; w < WAVES; ++w) M = fmaxf(M, sm_ml[w][0]);
        float L = 0.f, od = 0.f;
        for (int w = 0; w < WAVES; ++w) {
            const float mw = sm_ml[w][0];
            const float f = (mw == -INFINITY) ? 0.f : expf(mw - M);
            L += sm_ml[w][1] * f; od += sm_o[w][d] * f;
        }
        if (a.out) {   // nsplit == 1: finished output, [H*D]
            a.out[(size_t)qi * E + h * D + d] = od / L;
        } else {       // split partial: combined by the consumer's prologue (ProAttn) or by attn_combine_kernel
            a.part_o[(size_t)(h * a.nsplit + s) * D + d] = od;
            if (d == 0) { a.part_ml[(h * a.nsplit + s) * 2] = M; a.part_ml[(h * a.nsplit + s) * 2 + 1] = L; }
        }
    }
    THK_STAMP(a.trace, bid, 3);
}

template <int D, int WAVES, bool KVH>
__global__ __launch_bounds__(WAVES * 64) void attn_decode_kernel(const AttnArgs a) {
    attn_body<D, WAVES, KVH>(a, blockIdx.x);
}

hipError_t launch_attn_decode(const AttnArgs& a, hipStream_t st) {
    const int grid = a.H * a.nsplit * (a.nq > 1 ? a.nq : 1);
    const bool w8 = a.waves == 8;
#define THK_ATTN(d)                                                                                           \
    case d:                                                                                                   \
        if (a.kv_f16) {                                                                                        \
            if (w8) hipLaunchKernelGGL((attn_decode_kernel<d, 8, true>), dim3(grid), dim3(512), 0, st, a);    \
            else hipLaunchKernelGGL((attn_decode_kernel<d, 4, true>), dim3(grid), dim3(256), 0, st, a);       \
        } else if (w8) hipLaunchKernelGGL((attn_decode_kernel<d, 8, false>), dim3(grid), dim3(512), 0, st, a); \
        else hipLaunchKernelGGL((attn_decode_kernel<d, 4, false>), dim3(grid), dim3(256), 0, st, a);          \
        break;
    switch (a.D) {
        THK_ATTN(64) THK_ATTN(128) THK_ATTN(256)
        default: return hipErrorInvalidValue;
    }
#undef THK_ATTN
    return hipGetLastError();
}

// Stand-alone combine of split partials -> out[H*D] (used by thk_attn_decode when nsplit > 1).
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
__global__ __launch_bounds__(kBlock) void finish_token_kernel(const unsigned long long* __restrict__ block_best, int nblocks,
                                                              SeqState* st, int32_t* gen_log, int log_cap,
                                                              const int* advance_ptr, int32_t* id_out, int n_ctx, unsigned* epoch,
                                                              unsigned long long* trace, unsigned long long* clock_log) {
    __shared__ unsigned long long sm[kBlock];
    THK_STAMP(trace, 0, 0);
    unsigned long long b = 0ull;
    for (int i = threadIdx.x; i < nblocks; i += kBlock) { const unsigned long long k = block_best[i]; b = k > b ? k : b; }
    sm[threadIdx.x] = b;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) { const unsigned long long o = sm[threadIdx.x + s]; if (o > sm[threadIdx.x]) sm[threadIdx.x] = o; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int32_t tok = (int32_t)(0xFFFFFFFFu - (unsigned)(sm[0] & 0xFFFFFFFFull));
        if (id_out) *id_out = tok;
        if (st) {
            st->token = tok;
            if (gen_log && st->n_gen < log_cap) {
                gen_log[st->n_gen] = tok;
                if (clock_log) clock_log[st->n_gen] = __builtin_amdgcn_s_memrealtime();   // 100 MHz chip-wide counter: when this step finished
            }
            st->n_gen += 1;
            if (advance_ptr && *advance_ptr && (n_ctx <= 0 || st->pos + 1 < n_ctx)) st->pos += 1;
        }
        if (epoch) *epoch += 1u;
    }
    THK_STAMP(trace, 0, 3);
}
hipError_t launch_finish_token(const unsigned long long* block_best, int nblocks, SeqState* st_dev, int32_t* gen_log, int log_cap,
                               const int* advance_ptr, int32_t* id_out, int n_ctx, unsigned* epoch, hipStream_t st, unsigned long long* trace,
                               unsigned long long* clock_log) {
    hipLaunchKernelGGL(finish_token_kernel, dim3(1), dim3(kBlock), 0, st, block_best, nblocks, st_dev, gen_log, log_cap, advance_ptr, id_out, n_ctx, epoch, trace, clock_log);
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
// ascending index - a total order, so the selection and its order are unique.  ONE workgroup of 1024 threads: every thread holds
// up to kTopkPer keys in registers; the k-th largest key is found by an 8-pass radix select on the 64-bit keys (256-bin LDS
// histogram of the next byte among the keys that match the prefix so far), the keys >= it are compacted into LDS (exactly k: keys
// are unique) and sorted with a bitonic network.  V <= 1024 * kTopkPer = 32768, k <= 1024.
constexpr int kTopkThreads = 1024, kTopkPer = 32, kTopkMax = 1024;     // 32 keys = 64 VGPRs per thread (16 waves per workgroup leave 128)
__global__ __launch_bounds__(kTopkThreads) void topk_kernel(const float* __restrict__ logits, int V, int k, unsigned long long* __restrict__ keys_out) {
    __shared__ unsigned hist[256];
    __shared__ unsigned long long sel[kTopkMax];
    __shared__ unsigned long long s_prefix;
    __shared__ unsigned s_need, s_count;
    const int tid = threadIdx.x;
    unsigned long long key[kTopkPer];
#pragma unroll
    for (int j = 0; j < kTopkPer; ++j) {
        const int i = tid + j * kTopkThreads;
        key[j] = i < V ? argmax_key(logits[i], (unsigned)i) : 0ull;      // 0 is below every real key (a real key has ~idx != 0 in its low word or a non-zero value word)
    }
    if (tid == 0) { s_prefix = 0ull; s_need = (unsigned)k; s_count = 0u; }
    __syncthreads();
    for (int pass = 7; pass >= 0; --pass) {                                  // most significant byte first
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        const unsigned long long hi_mask = pass == 7 ? 0ull : (~0ull << ((pass + 1) * 8));
#pragma unroll
        for (int j = 0; j < kTopkPer; ++j) {
            const int i = tid + j * kTopkThreads;
            if (i < V && (key[j] & hi_mask) == prefix) atomicAdd(&hist[(unsigned)(key[j] >> (pass * 8)) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {                                                      // walk the bins from the top until the k-th largest falls into one
            unsigned need = s_need, b = 255;
            for (;; --b) { if (hist[b] >= need || b == 0) break; need -= hist[b]; }
            s_need = need;
            s_prefix = prefix | ((unsigned long long)b << (pass * 8));
        }
        __syncthreads();
    }
    const unsigned long long kth = s_prefix;                                 // the k-th largest key itself
#pragma unroll
    for (int j = 0; j < kTopkPer; ++j) {
        const int i = tid + j * kTopkThreads;
        if (i < V && key[j] >= kth) { const unsigned at = atomicAdd(&s_count, 1u); if (at < (unsigned)kTopkMax) sel[at] = key[j]; }
    }
    __syncthreads();
    int n = 1;
    while (n < k) n <<= 1;                                                   // bitonic sort of n >= k slots, descending; empty slots are 0
    for (int i = tid; i < n; i += kTopkThreads) if (i >= k) sel[i] = 0ull;
    __syncthreads();
    for (int size = 2; size <= n; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < n; i += kTopkThreads) {
                const int p = i ^ stride;
                if (p > i) {
                    const bool desc = (i & size) == 0;
                    const unsigned long long a = sel[i], b = sel[p];
                    if ((a < b) == desc) { sel[i] = b; sel[p] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < k; i += kTopkThreads) keys_out[i] = sel[i];
}
hipError_t launch_topk(const float* logits, int V, int k, unsigned long long* keys_out, hipStream_t st) {
    if (V < 1 || V > kTopkThreads * kTopkPer || k < 1 || k > kTopkMax || k > V) return hipErrorInvalidValue;
    hipLaunchKernelGGL(topk_kernel, dim3(1), dim3(kTopkThreads), 0, st, logits, V, k, keys_out);
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
