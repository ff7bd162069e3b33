// thk_device.hpp — device-side helpers shared by the launch-path kernels (thk_kernels.hip) and the persistent engine
// (thk_engine_body.inc): wave64 DPP reductions, the f16 x f32 dot product of one 16-byte weight vector, arg-max keys.
// Internal; included only by .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "thk_kernels.hpp"

namespace thk {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- wave reductions
// DPP butterfly inside each row of 16 lanes (quad_perm, row_half_mirror,
// row_mirror), then the four row totals are combined through readlane.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);  // row_half_mirror
    v += dpp_f<0x140>(v);  // row_mirror
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}
__device__ __forceinline__ float rdlane(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// Full-wave sum; result is wave-uniform.
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    return (rdlane(v, 0) + rdlane(v, 16)) + (rdlane(v, 32) + rdlane(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(rdlane(v, 0), rdlane(v, 16)), fmaxf(rdlane(v, 32), rdlane(v, 48)));
}
// Sum over aligned groups of G lanes (G = 16, 32 or 64); every lane of a group gets its group's sum.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
    v = row16_sum(v);
    if (G >= 32) v += __shfl_xor(v, 16);
    if (G >= 64) v += __shfl_xor(v, 32);
    return v;
}

// eight f16 weights (hardware-decoded, v_cvt_f32_f16 == the reference's bit-trick decode, th.cpp:363-394) times eight f32 activations
__device__ __forceinline__ float dot8(h8 w, f4 xl, f4 xh, float acc) {
    acc = fmaf((float)w[0], xl.x, acc); acc = fmaf((float)w[1], xl.y, acc);
    acc = fmaf((float)w[2], xl.z, acc); acc = fmaf((float)w[3], xl.w, acc);
    acc = fmaf((float)w[4], xh.x, acc); acc = fmaf((float)w[5], xh.y, acc);
    acc = fmaf((float)w[6], xh.z, acc); acc = fmaf((float)w[7], xh.w, acc);
    return acc;
}

// greedy pick (th-llama.cpp:826-838): order-preserving key, ties go to the smaller index
__device__ __forceinline__ unsigned long long argmax_key(float v, unsigned idx) {
    unsigned b = __builtin_bit_cast(unsigned, v);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);          // order-preserving map
    return ((unsigned long long)b << 32) | (unsigned long long)(0xFFFFFFFFu - idx);   // ties: smaller idx wins
}

// ---------------------------------------------------------------- agent-coherent accesses (overlapped dispatch, thk_ovl.cpp)
// A kernel of the overlapped dispatch starts while its predecessor is still running, so whatever crosses from one launch to the
// next cannot rely on the kernel-boundary cache maintenance: MI355X has one L2 per XCD and they are not coherent with each
// other.  COH = true turns an access into its agent-scope form (sc1: loads are served from the memory side, stores write
// through); COH = false is the plain access of the stream-ordered kernels.
// Vector forms use raw buffer instructions (base = wave-uniform pointer in SGPRs, 32-bit byte offset per lane): one instruction
// per 16 bytes, cache-policy bits in the instruction (aux: 16 = sc1, 2 = nt).
// (Results are converted with WHOLE-vector bit casts: __builtin_bit_cast applied to one element of an ext_vector - v.y, t[1] -
// reads element 0 with this toolchain, ROCm 7.2 clang; found the hard way.)
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
constexpr int kAuxSc1 = 16, kAuxNt = 2;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t coh_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFF0, 0x00020000);
}
template <bool COH, int AUX = kAuxSc1>
__device__ __forceinline__ f4 ld_f4(const float* base, int elem) {
    if (!COH) return *reinterpret_cast<const f4*>(base + elem);
    const u4v t = __builtin_amdgcn_raw_buffer_load_b128(coh_rsrc(base), elem * 4, 0, AUX);
    return __builtin_bit_cast(f4, t);
}
template <bool COH>
__device__ __forceinline__ float2 ld_f2(const float* base, int elem) {
    if (!COH) return *reinterpret_cast<const float2*>(base + elem);
    const u2v t = __builtin_amdgcn_raw_buffer_load_b64(coh_rsrc(base), elem * 4, 0, kAuxSc1);
    const f2v f = __builtin_bit_cast(f2v, t);
    return float2{f.x, f.y};
}
template <bool COH>
__device__ __forceinline__ float ld_f1(const float* p) {
    if (!COH) return *p;
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool COH>
__device__ __forceinline__ void st_f1(float* p, float v) {
    if (!COH) *p = v; else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool COH>
__device__ __forceinline__ void st_h1(_Float16* p, _Float16 v) {
    if (!COH) *p = v;
    else __hip_atomic_store(reinterpret_cast<unsigned short*>(p), __builtin_bit_cast(unsigned short, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool COH>
__device__ __forceinline__ void st_f4(float* base, int elem, f4 v) {
    if (!COH) { *reinterpret_cast<f4*>(base + elem) = v; return; }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), coh_rsrc(base), elem * 4, 0, kAuxSc1);
}
template <bool COH>
__device__ __forceinline__ void st_u64(unsigned long long* p, unsigned long long v) {
    if (!COH) *p = v; else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The dependency protocol of the overlapped dispatch.  Every launch owns kOvlShards arrival counters, 128 bytes apart; a
// workgroup that has drained its write-through stores adds its wave count to shard (workgroup % kOvlShards).  A successor's
// workgroup waits - after it has requested its first batch of weights, which depend on nothing - until the shards add up to the
// predecessor's wave count: wave 0 polls (one shard per lane, summed with a butterfly), the other waves sit at the barrier.
// The counters are zeroed by the step's last launch.  A wait is bounded (0.5 s of the 100 MHz clock); the first one that expires
// sets *err, and every later wait of the chain returns at once, so a broken chain ends in well under a second per step.
__device__ __forceinline__ void ovl_wait(const OvlLink& L) {
    if ((threadIdx.x >> 6) == 0 && L.wait) {
        const int lane = threadIdx.x & 63;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            unsigned v = lane < kOvlShards ? __hip_atomic_load(L.wait + lane * kOvlShardWords, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                           : (lane == kOvlShards ? __hip_atomic_load(L.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 24 : 0u);
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (v >= L.wait_n) break;                                  // all arrived (or the chain is already broken: err << 24)
            __builtin_amdgcn_s_sleep(8);                              // ~0.25 us between polls; s_sleep(2) measured 0.4 % slower
            if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ull) {
                if (lane == 0) __hip_atomic_store(L.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}
// Counters count WAVES (wait_n = workgroups x waves per workgroup); thread 0 arrives for the whole workgroup behind a barrier.
// (Every wave arriving for itself, without the barrier, measured 0.2 % slower.)
template <int WPB, bool BLOCK>
__device__ __forceinline__ void ovl_arrive(const OvlLink& L, int bid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every wave: its write-through stores have left
    __syncthreads();
    if (threadIdx.x == 0 && L.done)
        (void)__hip_atomic_fetch_add(L.done + (bid & (kOvlShards - 1)) * kOvlShardWords, (unsigned)WPB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace thk
