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

// ---------------------------------------------------------------- agent-scope accesses
// MI355X has one L2 per XCD and they are not coherent with each other: what one workgroup hands to another INSIDE a launch has to
// cross through memory.  An agent-scope store is written through (sc1), an agent-scope load is served from the memory side, an
// agent-scope atomic is performed there.  Used by the one in-launch hand-off of the decode step: the lm-head's last workgroup
// picks the greedy token from every workgroup's arg-max key (gemv_body, EPI_HEAD).
__device__ __forceinline__ void st_agent_u64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_agent_u64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace thk
