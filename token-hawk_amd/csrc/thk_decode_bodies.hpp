// thk_decode_bodies.hpp — device bodies of the decode step's kernels: the fused mat-vec with its prologues / epilogues (the lm-head
// flavour also finishes the token), attention, the stand-alone greedy pick.  Launched by thk_kernels.hip (stream launches, hipGraph
// replays).  Internal; device code only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
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

template <int NT, bool AGENT>
__device__ __forceinline__ void finish_token_reduce(const FinishArgs& a, unsigned long long* keys, int nkeys, unsigned long long* sm);

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
//   pin()    - the loaded activation values pass through empty asm statements (the first one clobbers "memory"): the prologue's
//              arithmetic cannot be placed ahead of the weight requests any more (the compiler did exactly that to the split
//              combine once its address arithmetic had become cheap: weights requested 525 instructions into the kernel)
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
    __device__ __forceinline__ void pin() {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < KP; ++k) asm volatile("" : "+v"(v[k]));
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
    __device__ __forceinline__ void pin() {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < KP; ++k) { asm volatile("" : "+v"(v[k])); asm volatile("" : "+v"(g[k])); }
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
    // contig: the thread's KP float4 are CONSECUTIVE (thread t owns #(t * KP + k)) and lie in one head - true for every head size the
    // attention kernels accept (64 | 128 | 256) when 4 KP divides 64.  One (m, l) set per thread instead of KP, and the merge's exp /
    // reciprocal are computed once (same arithmetic, same order - the compiler shares them across the KP calls of attn_merge);
    // otherwise float4 #(t + k * BT) as the other prologues.  (Compile-time: a run-time branch around the requests made the
    // compiler wait for vmcnt(0) at the join, ahead of the weight requests.)
    static constexpr bool contig = (64 % (4 * KP)) == 0;
    __device__ __forceinline__ void load1(const GemvArgs& a, int e, float2 (&m)[NSP], f4 (&o)[NSP]) {
        // element e -> (head, offset) by a multiply-high with d_magic = ceil(2^32 / D), exact for e, D < 65536 (launch_gemv sets it and
        // checks the range; no division fallback - the compiler would compute it speculatively).  The division was ~130 instructions
        // between the prologue's requests and a.D a scalar load to wait for: 0.5 us per launch, measured.
        const int h = (int)__umulhi((unsigned)e, a.d_magic), d = e - h * a.D;
#pragma unroll
        for (int s = 0; s < NSP; ++s) {
            m[s] = *reinterpret_cast<const float2*>(a.part_ml + (h * NSP + s) * 2);
            o[s] = *reinterpret_cast<const f4*>(a.part_o + (h * NSP + s) * a.D + d);
        }
    }
    __device__ __forceinline__ int f4_index(int k) const { return contig ? (int)threadIdx.x * KP + k : (int)threadIdx.x + k * BT; }
    __device__ __forceinline__ void issue(const GemvArgs& a) {
        if (NS == 0) return;
        if constexpr (contig) {
            const int e0 = min((int)threadIdx.x * KP * 4, a.C - 4 * KP);
            const int h = (int)__umulhi((unsigned)e0, a.d_magic), d = e0 - h * a.D;
#pragma unroll
            for (int s = 0; s < NSP; ++s) {
                ml[0][s] = *reinterpret_cast<const float2*>(a.part_ml + (h * NSP + s) * 2);
#pragma unroll
                for (int k = 0; k < KP; ++k) ov[k][s] = *reinterpret_cast<const f4*>(a.part_o + (h * NSP + s) * a.D + d + 4 * k);
            }
#pragma unroll
            for (int k = 1; k < KP; ++k)
#pragma unroll
                for (int s = 0; s < NSP; ++s) ml[k][s] = float2{0.f, 0.f};
        } else {
#pragma unroll
            for (int k = 0; k < KP; ++k) load1(a, min((int)(threadIdx.x + k * BT) << 2, a.C - 4), ml[k], ov[k]);
        }
    }
    __device__ __forceinline__ void pin() {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < KP; ++k)
#pragma unroll
            for (int s = 0; s < NSP; ++s) { asm volatile("" : "+v"(ov[k][s])); asm volatile("" : "+v"(ml[k][s].x), "+v"(ml[k][s].y)); }
    }
    __device__ __forceinline__ void finish(const GemvArgs& a, float* xs, float*, int ns, int) {
        const int C = a.C;
        if (NS != 0) {
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int i = f4_index(k);
                f4 res = contig ? attn_merge<NSP>(ml[0], ov[k]) : attn_merge<NSP>(ml[k], ov[k]);
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

    // EPI_ROPE_KV: which of wq | wk | wv row r0 of the virtual [3E, E] stack belongs to.  Two compares instead of a division (the
    // division sat ahead of the kernel's first weight request); the readfirstlane keeps the result opaque - once the optimiser
    // knows it is 0, 1 or 2 it turns `which == 0 ? a.W[0] : ...` into the dynamically indexed a.W[which] and the whole argument
    // block moves to scratch memory (seen in the ISA: 304 bytes of private segment, every member reloaded from it).
    auto matrix_of = [&](int r0) -> int { return __builtin_amdgcn_readfirstlane((r0 >= a.E) + (r0 >= 2 * a.E)); };
    // row pointers of group g (wave-uniform; independent of the activation vector)
    auto row_ptrs = [&](int g, const h8* (&rp)[NR]) {
        if (EPI == EPI_ROPE_KV) {
            const int r0 = NR == 1 ? g : 2 * g, which = matrix_of(r0), rr = r0 - which * a.E;      // (no division ahead of the first weight request; NR == 1: single rows, below)
            const uint16_t* base = which == 0 ? a.W[0] : (which == 1 ? a.W[1] : a.W[2]);
            rp[0] = reinterpret_cast<const h8*>(base + (size_t)rr * C);
            if (NR > 1) rp[1 % NR] = reinterpret_cast<const h8*>(base + (size_t)(rr + 1) * C);
        } else if (EPI == EPI_SWIGLU && NR == 1) {     // single rows: group 2 j is row j of w1, group 2 j + 1 row j of w3 (the pair meets in LDS, below)
            // (not `g & 1 ? a.W[1] : a.W[0]`: a select of two elements of one array becomes the dynamically indexed a.W[g & 1] and the
            // whole argument block moves to scratch memory - seen in the ISA)
            // (the distance in BYTES on integers: w1 and w3 may live in different allocations, where a typed pointer difference is undefined
            // and would drop an odd byte; advisor, round 4)
            // The address stays an offset from the kernel argument a.W[0] (a pointer rebuilt from an integer loses its address space: the
            // weight stream turned into FLAT loads and the launch took 33.5 instead of 30.2 us, measured in round 5).
            const ptrdiff_t to_w3 = (g & 1) ? (ptrdiff_t)(reinterpret_cast<uintptr_t>(a.W[1]) - reinterpret_cast<uintptr_t>(a.W[0])) : (ptrdiff_t)0;
            rp[0] = reinterpret_cast<const h8*>(reinterpret_cast<const char*>(a.W[0]) + to_w3 + (size_t)(g >> 1) * C * 2);
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
    int pos_pipe = 0;        // PIPE && EPI_ROPE_KV: the position, read behind the first weight batch (below) - its null check is a branch that
                             // would otherwise wait for a scalar load of a struct member before anything has been requested
    auto head_offset = [&](int rr) -> int { return rr % a.D; };
    // Single rows for the epilogues that need two row sums together (NR == 1; round 4).  EPI_ROPE_KV: a wave owns whole rows w, w + W, ... (8 KiB chunks instead of the 16 KiB of a
    // row pair - tools/probes/read_floor_probe.hip: the same bytes stream 4-5 % faster in 8 KiB chunks).  The rotation needs rows
    // 2j and 2j + 1, which then sit in NEIGHBOURING waves of one workgroup (waves 0|1 and 2|3; W is a multiple of 4): the sums go
    // to LDS (ysm[round][wave]) and thread t < 2 * rounds rotates and stores pair (round t >> 1, waves 2 (t & 1), 2 (t & 1) + 1) after
    // one barrier at the end of the kernel; its cos/sin are requested right behind the first weight batch.  EPI_SWIGLU the same way:
    // group 2j = row j of w1, group 2j + 1 = row j of w3, thread t writes silu(u1) * u3 for pair (round t >> 1, waves 2 (t & 1) | + 1).
    constexpr int kSrRounds = 32;                       // rounds a wave may run (launch_gemv checks the geometry)
    float* ysm = red + 32;                              // [kSrRounds][WPB]
    int sr_count = 0;
    int sr_r0 = -1, sr_which = 0, sr_rr = 0;
    float2 sr_cs = {1.f, 0.f};
    auto epi_fetch = [&](int g, EpiOps& eo) {
        if (EPI == EPI_RESID) {
#pragma unroll
            for (int r = 0; r < NR; ++r) eo.resid[r] = a.resid[min(NR * g + r, a.R - 1)];      // wave-uniform address: one request
        } else if (EPI == EPI_ROPE_KV && NR > 1) {
            const int r0 = 2 * g, which = matrix_of(r0), rr = r0 - which * a.E, j = head_offset(rr);
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
        } else if (EPI == EPI_ROPE_KV && NR == 1) {     // single rows: the sum waits in LDS for its RoPE partner (the neighbouring wave's row)
            if (lane == 0) ysm[sr_count * WPB + wave] = acc[0];
            ++sr_count;
        } else if (EPI == EPI_ROPE_KV) {     // K6 th.cpp:1476-1490 + K/V append th-llama.cpp:332-339
            if (lane == 0) {
                const int pos = PIPE ? pos_pipe : (a.pos_ptr ? *a.pos_ptr : a.pos_val);
                const int r0 = 2 * g, which = matrix_of(r0), rr = r0 - which * a.E;
                float y0 = acc[0], y1 = acc[1 % NR];
                if (which < 2) {
                    const int j = head_offset(rr);    // even
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
        } else if (EPI == EPI_SWIGLU && NR == 1) {      // single rows: the sum waits in LDS for its partner (row j of the other matrix, the neighbouring wave)
            if (lane == 0) ysm[sr_count * WPB + wave] = acc[0];
            ++sr_count;
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
    // EPI_ROPE_KV: the position is requested with the kernel's first instruction (its pointer arrives preloaded, gemv_kernel) and
    // first needed behind finish(): by then the scalar load has returned and finish()'s own lgkmcnt waits (LDS) find nothing pending
    if (EPI == EPI_ROPE_KV && (PIPE || NR == 1)) { pos_pipe = *a.pos_ptr; __builtin_amdgcn_sched_barrier(0); }
    // --- memory traffic is ordered: activation loads, then the wave's first weight batch (weights
    // do not depend on the activations), then the prologue math while the weights are in flight.
    int g = wave_global;
    const bool has_first = g < a.n_groups;
    const h8* rp0[NR];
    h8 w0[NR][U];
    typename ProSelect<NS, PRO, NSP, WPB>::type pro;
    pro.issue(a);
    __builtin_amdgcn_sched_barrier(0);
    row_ptrs(has_first ? g : a.n_groups - 1, rp0);   // idle waves (more waves than groups) load a valid row: an unconditional load keeps
                                                     // the vmcnt bookkeeping exact.  (Behind issue(): the matrix select of the qkv stack is
                                                     // ~40 instructions the activation requests need not wait for.)
    load_batch(rp0, 0, w0);
    __builtin_amdgcn_sched_barrier(0);
    if (NS != 0 && PRO == PRO_ATTN) pro.pin();       // (the other prologues keep their order without it, and lose ~30 instructions of head start with it)
    pro.finish(a, xs, red, ns, bid);
    if (EPI == EPI_ROPE_KV && NR == 1) {
        static_assert(EPI != EPI_ROPE_KV || NR != 1 || WPB == 4, "RoPE pairs are waves 0|1 and 2|3 of a 4-wave workgroup");
        const int t = threadIdx.x, r0 = bid * WPB + 2 * (t & 1) + (t >> 1) * total_waves;      // even: bid * 4 and total_waves are
        const bool mine = t < 2 * kSrRounds && r0 < a.n_groups;
        const int rc = mine ? r0 : 0;                                                           // branch-free request (a per-lane `if` around a load costs a vmcnt(0))
        sr_which = (rc >= a.E) + (rc >= 2 * a.E); sr_rr = rc - sr_which * a.E;
        const int j = head_offset(sr_rr);
        sr_cs = *reinterpret_cast<const float2*>(a.rope_tab + ((size_t)pos_pipe * (a.D >> 1) + (j >> 1)) * 2);
        sr_r0 = mine ? r0 : -1;
    }
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
    if (EPI == EPI_SWIGLU && NR == 1) {       // the (w1, w3) pairs of this workgroup (see ysm above)
        static_assert(EPI != EPI_SWIGLU || NR != 1 || WPB == 4, "pairs are waves 0|1 and 2|3 of a 4-wave workgroup");
        __syncthreads();
        const int t = threadIdx.x, u0 = bid * WPB + 2 * (t & 1) + (t >> 1) * total_waves;       // even
        if (t < 2 * kSrRounds && u0 < a.n_groups) {
            const float u1 = ysm[(t >> 1) * WPB + 2 * (t & 1)], u3 = ysm[(t >> 1) * WPB + 2 * (t & 1) + 1];
            a.y[u0 >> 1] = (u1 / (1.0f + expf(-u1))) * u3;          // K12 th.cpp:2706-2707, K13 :2512-2524
        }
    }
    if (EPI == EPI_ROPE_KV && NR == 1) {      // the RoPE pairs of this workgroup (see ysm above)
        __syncthreads();
        if (sr_r0 >= 0) {
            const int t = threadIdx.x;
            float y0 = ysm[(t >> 1) * WPB + 2 * (t & 1)], y1 = ysm[(t >> 1) * WPB + 2 * (t & 1) + 1];
            if (sr_which < 2) {                                     // K6 th.cpp:1476-1490
                const float t0 = y0 * sr_cs.x - y1 * sr_cs.y, t1 = y0 * sr_cs.y + y1 * sr_cs.x;
                y0 = t0; y1 = t1;
            }
            if (sr_which != 0 && a.kv_f16) {                        // K/V append th-llama.cpp:332-339 (optional f16 cache: RNE at the append)
                _Float16* dh = reinterpret_cast<_Float16*>(sr_which == 1 ? a.kcache : a.vcache) + (size_t)pos_pipe * a.E;
                dh[sr_rr] = (_Float16)y0; dh[sr_rr + 1] = (_Float16)y1;
            } else {
                float* dst = sr_which == 0 ? a.y : (sr_which == 1 ? a.kcache + (size_t)pos_pipe * a.E : a.vcache + (size_t)pos_pipe * a.E);
                dst[sr_rr] = y0; dst[sr_rr + 1] = y1;
            }
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
            // folded pick: the key crosses to another XCD inside this launch, so it is written through; nobody waits for it here
            if (a.fin.folded) st_agent_u64(a.block_best + bid, b); else a.block_best[bid] = b;
        }
        // The greedy pick folded into this launch (round 4).  ONE workgroup waits until every key slot is non-zero (a key is never 0
        // and the slots are zeroed again below), reduces them and finishes the token.  The other workgroups pay one fire-and-forget
        // store: a first version in which every workgroup waited for its store's acknowledgement and then drew a ticket with a
        // returning atomic lengthened the launch by 8 us (two memory round trips per workgroup on 1.6 generations of workgroups) -
        // more than the finish_token launch and its boundary (4.3 + 2.0 us) cost.
        // Forward progress does NOT depend on the order workgroups are dispatched in: the waiting workgroup holds one workgroup slot
        // of one CU, every other slot of the device keeps taking the launch's remaining workgroups, so the keys arrive wherever
        // the poller was placed (needs: the device can host two workgroups of this launch at once - true for any CU).  The
        // dispatch order only decides how long the poller spins: folded == 1 picks the HIGHEST index (normally dispatched last:
        // it spins for a few hundred ns), folded == 2 picks workgroup 0 (normally dispatched FIRST: it spins through the whole
        // launch) - kept as a tested mode exactly to show that the result does not lean on the order (tunable fold_finish = 2).
        if (a.fin.folded && bid == (a.fin.folded == 2 ? 0 : nblk - 1)) {
            __syncthreads();                     // wave 0's key store is issued; xs (the activation vector) is dead from here on
            finish_token_reduce<WPB * 64, true>(a.fin, a.block_best, nblk, reinterpret_cast<unsigned long long*>(xs));
        }
    }
}
// ---------------------------------------------------------------- quarter-row mat-vec (w2: y = resid + W u, rows of 21.5 / 27 KiB)
// Round 4.  tools/probes/read_floor_probe.hip: a launch streams 6-8 % faster when a wave's contiguous chunk is 4-8 KiB than when it
// is a whole w2 row (21.5 KiB for F = 11008).  Here a WORKGROUP owns a row - its four waves take a quarter each (5.5 KiB; quarters
// are whole half-slots of 32 vectors, the last one is the short one) - and workgroup b takes rows b, b + B, ...: the launch sweeps
// the matrix front to back in 4-5 KiB pieces.  The quarter sums meet in LDS (qsm[round][wave]) and thread t adds row t's four and the
// residual after ONE barrier at the end of the kernel (residuals requested at the start).  NR rows are in flight per wave (a ring of
// NR * NSQ 16-byte loads per lane, slot c of row k + NR requested right behind the multiply of slot c of row k).
// NS = slot class of the column count (22 | 27 | 8 | 10), NSQ = slots a quarter spans (its last slot may be half or less).
template <int NS, int NR, int WPB = kWaves>
__device__ __forceinline__ void gemv_quarter_body(const GemvArgs& a, const int bid, const int nblk) {
    static_assert(WPB == 4 && NS != 0, "a row is cut into one quarter per wave of a 4-wave workgroup");
    constexpr int QV = ((NS * 64 + 3) / 4 + 31) / 32 * 32;        // vectors (8 columns) per quarter: whole half-slots
    constexpr int NSQ = (QV + 63) / 64;
    constexpr int kRounds = 32;                                  // rows a workgroup may own (launch_gemv_quarter checks)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int C = a.C, nvec = C >> 3;
    float* xs = smem;
    float* red = smem + (NS << 9);
    float* qsm = red + 32;                                        // [kRounds][WPB]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const f4* xlo = reinterpret_cast<const f4*>(xs);
    const f4* xhi = reinterpret_cast<const f4*>(xs + (NS << 8));
    const int v0 = wave * QV, v1 = min(v0 + QV, nvec);           // this wave's vectors of every row
    THK_STAMP(a.trace, bid, 0);
    ProCopy<NS, WPB> pro;
    pro.issue(a);
    __builtin_amdgcn_sched_barrier(0);
    const int rows = a.R;
    const int t_row = bid + (int)threadIdx.x * nblk;             // the row thread t finishes (t < rounds)
    const float my_resid = a.resid[min(t_row, rows - 1)];
    const uint16_t* W = a.W[0];
    auto row_ptr = [&](int r) -> const h8* { return reinterpret_cast<const h8*>(W + (size_t)min(r, rows - 1) * C); };
    h8 ring[NR][NSQ];
    auto slot_vec = [&](int c) -> int { return v0 + c * 64 + lane; };
    auto load_slot = [&](const h8* rp, int c, h8& dst) { dst = __builtin_nontemporal_load(rp + min(slot_vec(c), v1 - 1)); };
    int r = bid;                                                 // row of ring position 0
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const h8* rp = row_ptr(r + k * nblk);
#pragma unroll
        for (int c = 0; c < NSQ; ++c) load_slot(rp, c, ring[k][c]);
    }
    __builtin_amdgcn_sched_barrier(0);
    pro.finish(a, xs, red, NS, bid);
    THK_STAMP(a.trace, bid, 1);
    int round = 0;
    for (; r < rows; r += NR * nblk) {
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int rk = r + k * nblk;                         // wave-uniform
            // past the matrix the refills go to row 0 (the same 22 KiB for every workgroup: L2 hits); a branch around the requests would
            // cost a vmcnt(0) at its join
            const h8* rpn = rk + NR * nblk < rows ? row_ptr(rk + NR * nblk) : reinterpret_cast<const h8*>(W);
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < NSQ; ++c) {
                const int v = slot_vec(c);
                const f4 xl = xlo[min(v, (NS << 6) - 1)], xh = xhi[min(v, (NS << 6) - 1)];
                const float p = dot8(ring[k][c], xl, xh, 0.f);
                acc += v < v1 ? p : 0.f;                         // a quarter's last slot(s) reach past its end (the last quarter is the short one)
                __builtin_amdgcn_sched_barrier(0);
                load_slot(rpn, c, ring[k][c]);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc = wave_sum(acc);
            if (lane == 0 && rk < rows) qsm[(round + k) * WPB + wave] = acc;
        }
        round += NR;
    }
    THK_STAMP(a.trace, bid, 2);
    __syncthreads();
    if ((int)threadIdx.x < kRounds && t_row < rows) {
        const float* q = qsm + threadIdx.x * WPB;
        a.y[t_row] = my_resid + ((q[0] + q[1]) + (q[2] + q[3]));    // K11 th.cpp:2136-2147 on top of K1
    }
    THK_STAMP(a.trace, bid, 3);
}
// ---------------------------------------------------------------- attention (decode)
// grid = H * nsplit blocks; block (h, s) owns positions [s*tc, (s+1)*tc) of head h.
// A position's head slice is D contiguous elements; D/EPL lanes x 16 bytes cover it (EPL = 4 f32 elements, or 8 binary16 ones: KvLane / AttnGeo below),
// so a wave-instruction fetches PPW = 64/(D/EPL) positions of K (and of V).  Each wave runs an online softmax over its positions, the waves are merged through LDS, and the block
// writes (m, l, o[D]) for the split (or the normalised output when nsplit == 1).
// Scores: S = (q.k) * 1/sqrt(D) scaled after the sum (th.cpp:527-529, th-llama.cpp:518); softmax K10 th.cpp:1901-1957.
// WAVES waves per block share one (head, split): more waves = fewer positions per wave, so every wave needs a single load batch
// (one HBM round trip) at T = 512 with 4 splits.
// (Round 4 also had workgroup PAIRS that shared a split's K rows and halved its V columns - 256 workgroups for 7B without doubling
// wo's partials: null at T = 512, 50 % slower at T = 2048, the launch is a chain of dependent round trips and not bound by what one
// CU pulls; removed in round 5)
//
// tc_dyn (round 4): tc follows the LIVE context, tc = ceil(T / nsplit) rounded up to the wave batch (PPW * UB positions), computed
// here from the device-resident position - every (head, split) has work at every T >= nsplit * PPW * UB instead of the splits
// partitioning the cache capacity n_ctx (at T << n_ctx all but the first would exit empty).
// A lane's slice of one cached position: EPL consecutive elements = ONE 16-byte load whatever the cache type - f32 cache: 4 elements, D/4 lanes per
// position; binary16 cache (round 6): 8 elements, D/8 lanes per position, so a wave-instruction moves 1 KiB again (rounds 3-5 loaded 8 bytes per lane
// with the f32 lane mapping: 512 B per wave-instruction, twice the instructions in flight for the same bytes - 40 % of the HBM peak at T = 2048
// against the f32 cache's 56 %).  D = 64 with a binary16 cache keeps 4 elements per lane (group_sum needs groups of >= 16 lanes).
template <bool KVH, int EPL> struct KvLane;
template <> struct KvLane<false, 4> {
    typedef f4 raw;
    static __device__ __forceinline__ raw ld(const float* base, size_t elem_off) { return __builtin_nontemporal_load(reinterpret_cast<const f4*>(base + elem_off)); }
    static __device__ __forceinline__ float dot(const float (&q)[4], const raw& k) { return q[0] * k.x + q[1] * k.y + q[2] * k.z + q[3] * k.w; }
    static __device__ __forceinline__ void axpy(float (&o)[4], const raw& v, float p) { o[0] += v.x * p; o[1] += v.y * p; o[2] += v.z * p; o[3] += v.w * p; }
};
template <> struct KvLane<true, 4> {
    typedef f4 raw;       // widened at the load (v_cvt_f32_f16)
    static __device__ __forceinline__ raw ld(const float* base, size_t elem_off) {
        const h4 h = __builtin_nontemporal_load(reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(base) + elem_off));
        return f4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    }
    static __device__ __forceinline__ float dot(const float (&q)[4], const raw& k) { return q[0] * k.x + q[1] * k.y + q[2] * k.z + q[3] * k.w; }
    static __device__ __forceinline__ void axpy(float (&o)[4], const raw& v, float p) { o[0] += v.x * p; o[1] += v.y * p; o[2] += v.z * p; o[3] += v.w * p; }
};
template <> struct KvLane<true, 8> {
    typedef h8 raw;       // stays packed (4 VGPRs) until it is used
    static __device__ __forceinline__ raw ld(const float* base, size_t elem_off) {
        return __builtin_nontemporal_load(reinterpret_cast<const h8*>(reinterpret_cast<const _Float16*>(base) + elem_off));
    }
    static __device__ __forceinline__ float dot(const float (&q)[8], const raw& k) {
        return ((q[0] * (float)k[0] + q[1] * (float)k[1]) + (q[2] * (float)k[2] + q[3] * (float)k[3])) + ((q[4] * (float)k[4] + q[5] * (float)k[5]) + (q[6] * (float)k[6] + q[7] * (float)k[7]));
    }
    static __device__ __forceinline__ void axpy(float (&o)[8], const raw& v, float p) {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] += (float)v[i] * p;
    }
};
#ifndef THK_ATTN_UB8
#define THK_ATTN_UB8 4      // wave-instructions per K (and V) batch with 8-element lanes: PPW * UB = 16 positions per wave batch, as the f32 cache (every wave of the workgroup has work at T = 512 with 4 splits)
#endif
template <int D, bool KVH> struct AttnGeo {
    static constexpr int EPL = (KVH && D >= 128) ? 8 : 4;    // elements per lane
    static constexpr int LPP = D / EPL;                       // lanes per position
    static constexpr int PPW = 64 / LPP;                      // positions per wave-instruction
    static constexpr int UB = EPL == 8 ? THK_ATTN_UB8 : 8;    // wave-instructions per batch (K and V each)
};
template <int D, int WAVES, bool KVH, bool PIPE = false>
__device__ __forceinline__ void attn_body(const AttnArgs& a, const int bid) {
    typedef AttnGeo<D, KVH> G;
    constexpr int EPL = G::EPL, LPP = G::LPP, PPW = G::PPW, UB = G::UB;
    typedef KvLane<KVH, EPL> KL;
    typedef typename KL::raw kvraw;
    __shared__ float sm_o[WAVES][D];
    __shared__ float sm_ml[WAVES][2];

    const int pos0 = a.pos_ptr ? *a.pos_ptr : a.pos_val;        // requested first: a scalar load from memory whose latency the index arithmetic below hides
    __builtin_amdgcn_sched_barrier(0);
    // prefill: nq > 1 causal queries share one launch; query qi sits at position pos + qi
    const int hs = a.H * a.nsplit;
    const int qi = a.nq > 1 ? bid / hs : 0;
    const int hb = bid - qi * hs;
    // x / nsplit by multiply-high (ns_magic = ceil(2^32 / nsplit); 0 = nsplit is 1): an integer division is ~40 instructions, and two
    // of them stood between the kernel's entry and the request for the position every K/V address depends on
    auto div_ns = [&](int x) -> int { return a.ns_magic ? (int)__umulhi((unsigned)x, a.ns_magic) : x; };
    const int h = div_ns(hb), s = hb - h * a.nsplit;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / LPP, li = lane - grp * LPP;
    THK_STAMP(a.trace, bid, 0);
    const int T = pos0 + qi + 1;
    const int E = a.H * D;
    int tc = a.tc;
    if (a.tc_dyn) tc = ((div_ns(T + a.nsplit - 1) + PPW * UB - 1) / (PPW * UB)) * (PPW * UB);
    const int t0 = s * tc, t1 = min(t0 + tc, T);

    float q[EPL];
    {
        const f4 q0 = *reinterpret_cast<const f4*>(a.q + qi * E + h * D + li * EPL);
        q[0] = q0.x; q[1] = q0.y; q[2] = q0.z; q[3] = q0.w;
        if constexpr (EPL == 8) {
            const f4 q1 = *reinterpret_cast<const f4*>(a.q + qi * E + h * D + li * EPL + 4);
            q[4] = q1.x; q[5] = q1.y; q[6] = q1.z; q[7] = q1.w;
        }
    }
    const size_t koff = (size_t)(h * D + li * EPL);             // element offset of this lane's K (and V) slice inside a cache row

    float m = -INFINITY, l = 0.f;
    float o[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) o[i] = 0.f;
    // wave w takes positions t0 + (it*WAVES + w)*PPW*UB + u*PPW + grp.  Loads are branch-free:
    // positions past the end are clamped to a valid row and masked out of the softmax.
    // (Round 3 tried fetching the first batch BEFORE the device-resident position is known - every row below n_ctx is
    // allocated - to take the position's round trip off the critical path: 0.6 us per launch SLOWER on MI355X, removed.)
    auto fetch = [&](int tb, kvraw (&kv)[UB], kvraw (&vv)[UB]) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int t = min(tb + u * PPW + grp, t1 - 1);
            kv[u] = KL::ld(a.kcache, (size_t)t * E + koff);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int t = min(tb + u * PPW + grp, t1 - 1);
            vv[u] = KL::ld(a.vcache, (size_t)t * E + koff);
        }
    };
    // one batch: scores, online-softmax update, P.V
    auto consume = [&](int tb, const kvraw (&kv)[UB], const kvraw (&vv)[UB]) {
        float sc[UB];
        float bm = -INFINITY;
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int t = tb + u * PPW + grp;
            float d = KL::dot(q, kv[u]);
            d = group_sum<LPP>(d) * a.scale;
            sc[u] = (t < t1) ? d : -INFINITY;
            bm = fmaxf(bm, sc[u]);
        }
        bm = wave_max(bm);                      // wave-uniform, finite (tb < t1 => lane group 0 valid)
        const float mn = fmaxf(m, bm);
        const float alpha = (m == -INFINITY) ? 0.f : expf(m - mn);
        l *= alpha;
#pragma unroll
        for (int i = 0; i < EPL; ++i) o[i] *= alpha;
        float p[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) { p[u] = expf(sc[u] - mn); l += p[u]; }   // exp(-inf) == 0 for masked positions
#pragma unroll
        for (int u = 0; u < UB; ++u) KL::axpy(o, vv[u], p[u]);
        m = mn;
    };
    if constexpr (!PIPE) {
    for (int tb = t0 + wave * (PPW * UB); tb < t1; tb += WAVES * PPW * UB) {
        kvraw kv[UB], vv[UB];
        fetch(tb, kv, vv);
        __builtin_amdgcn_sched_barrier(0);
        consume(tb, kv, vv);
        THK_STAMP(a.trace, bid, 1);
    }
    } else {
    // PIPE (long caches: a (head, split) longer than one round of the workgroup, WAVES * PPW * UB positions): the rounds are software-
    // pipelined - the next round's K/V batch is requested BEFORE this round's softmax arithmetic, so a wave has two batches in
    // flight and the rounds are not a chain of dependent HBM round trips.  A variant of its own: the same structure cost the
    // single-round case (T <= 512 with 4 splits, the headline) 1 us per launch (profiles/r04_attention_ctx2048.txt).
    constexpr int STRIDE = WAVES * PPW * UB;
    int tb = t0 + wave * (PPW * UB);
    kvraw kv[UB], vv[UB];
    if (tb < t1) fetch(tb, kv, vv);
    while (tb < t1) {
        kvraw kvn[UB], vvn[UB];
        const int tn = tb + STRIDE;
        const bool more = tn < t1;                  // wave-uniform
        if (more) fetch(tn, kvn, vvn);
        __builtin_amdgcn_sched_barrier(0);
        consume(tb, kv, vv);
        THK_STAMP(a.trace, bid, 1);
        if (more) {
#pragma unroll
            for (int u = 0; u < UB; ++u) kv[u] = kvn[u];
#pragma unroll
            for (int u = 0; u < UB; ++u) vv[u] = vvn[u];
        }
        tb = tn;
    }
    }
    // merge the lane groups of the wave (same m)
#pragma unroll
    for (int off = LPP; off < 64; off <<= 1) {
        l += __shfl_xor(l, off);
#pragma unroll
        for (int i = 0; i < EPL; ++i) o[i] += __shfl_xor(o[i], off);
    }
    if (lane < LPP) {
        *reinterpret_cast<f4*>(&sm_o[wave][lane * EPL]) = f4{o[0], o[1], o[2], o[3]};
        if constexpr (EPL == 8) *reinterpret_cast<f4*>(&sm_o[wave][lane * EPL + 4]) = f4{o[4], o[5], o[6], o[7]};
    }
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

// ---------------------------------------------------------------- finishing a token
// Greedy pick + sequence bookkeeping once every workgroup of the head kernel has written its best key: reduce the keys, write the
// token (first max wins, th-llama.cpp:826-838), log it, advance the position when asked.  Runs as the tail of the lm-head launch's
// highest-numbered workgroup (gemv_body, EPI_HEAD with fin.folded; AGENT = the keys come from other XCDs inside the same launch,
// so every slot is polled until it is non-zero and zeroed again) or as a launch of its own (finish_token_kernel: engine path,
// tunable fold_finish = 0).
// n_ctx > 0: the position only advances while pos + 1 < n_ctx, so a decode loop that outruns the host-side check
// (thk_model_decode_step(s) refuse it) can never index the caches or the RoPE table out of bounds.
// epoch != NULL: the engine's tag epoch is bumped here, i.e. after the engine launch of this step and before the next.
template <int NT, bool AGENT>
__device__ __forceinline__ void finish_token_reduce(const FinishArgs& a, unsigned long long* keys, int nkeys, unsigned long long* sm /* NT u64 of LDS */) {
    unsigned long long b = 0ull;
    bool late = false;
    const unsigned long long t0 = AGENT ? __builtin_amdgcn_s_memrealtime() : 0ull;
    for (int i = threadIdx.x; i < nkeys; i += NT) {
        unsigned long long k;
        if (AGENT) {       // written by other workgroups of THIS launch: read from the memory side until it is there (bounded: 1 s of the 100 MHz clock)
            while ((k = ld_agent_u64(keys + i)) == 0ull) {
                if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) { late = true; break; }
                __builtin_amdgcn_s_sleep(4);
            }
            st_agent_u64(keys + i, 0ull);                // the slot is empty again for the next step's launch
        } else {
            k = keys[i];
        }
        b = k > b ? k : b;
    }
    if (AGENT && late && a.st) a.st->pad = 1;            // a workgroup of this launch never delivered: reported by thk_model_seq_get / seq_last_token
    sm[threadIdx.x] = b;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) { const unsigned long long o = sm[threadIdx.x + s]; if (o > sm[threadIdx.x]) sm[threadIdx.x] = o; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        SeqState* st = a.st;
        const int32_t tok = (int32_t)(0xFFFFFFFFu - (unsigned)(sm[0] & 0xFFFFFFFFull));
        if (a.id_out) *a.id_out = tok;
        if (st) {
            st->token = tok;
            if (a.gen_log && st->n_gen < a.log_cap) {
                a.gen_log[st->n_gen] = tok;
                if (a.clock_log) a.clock_log[st->n_gen] = __builtin_amdgcn_s_memrealtime();   // 100 MHz chip-wide counter: when this step finished
            }
            st->n_gen += 1;
            if (a.advance_ptr && *a.advance_ptr && (a.n_ctx <= 0 || st->pos + 1 < a.n_ctx)) st->pos += 1;
        }
        if (a.epoch) *a.epoch += 1u;
    }
}
__device__ __forceinline__ void finish_token_body(const FinishArgs& a) {
    __shared__ unsigned long long sm[kBlock];
    THK_STAMP(a.trace, 0, 0);
    finish_token_reduce<kBlock, false>(a, const_cast<unsigned long long*>(a.block_best), a.nblocks, sm);
    THK_STAMP(a.trace, 0, 3);
}

}  // namespace thk
