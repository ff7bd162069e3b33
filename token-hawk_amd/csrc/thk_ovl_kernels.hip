// thk_ovl_kernels.hip — the decode step's kernels as the overlapped dispatch runs them (thk_ovl.cpp): compiled to a code object
// of its own (libthk_ovl.hsaco, loaded with the HSA runtime) and dispatched on a private user-mode queue whose packets carry no
// barrier bit.  Same device bodies as thk_kernels.hip (thk_decode_bodies.hpp) with OVL = true: a workgroup requests its first
// weight batch, waits for its predecessor's arrival counters, reads what crosses the launch boundary agent-coherently, writes
// through, and arrives on its own counters.  Names encode the template arguments; thk_kernels.hip's launchers compose the same
// string when a step program is recorded, so a geometry that is not instantiated here is reported, not mis-launched.
// No gridDim / blockDim builtins anywhere in these kernels: the private queue does not fill the hidden kernel arguments.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 --genco thk_ovl_kernels.hip  (+ clang-offload-bundler --unbundle), see __graft_entry__.py
#include <hip/hip_runtime.h>
#include "thk_decode_bodies.hpp"

using namespace thk;

#define OVL_GEMV_F(NR, U, NS, PRO, EPI, NSP, PIPE, F)                                                                      \
    extern "C" __global__ __launch_bounds__(256) void thk_ovl_gemv_##NR##_##U##_##NS##_##PRO##_##EPI##_##NSP##_##PIPE##_f##F(const GemvArgs a) { \
        gemv_body<NR, U, NS, PRO, EPI, true, NSP, (PIPE) != 0, 4, OVL_QUEUE | F>(a, blockIdx.x, a.ovl.n_blocks);               \
    }
// four flavours of every instantiation: f0 behind a barrier packet and followed by one (plain accesses), f1 waits for its
// predecessor inside, f2 arrives for its successor, f3 both (thk_decode_bodies.hpp: OVL_WAIT = 1, OVL_ARRIVE = 2)
#define OVL_GEMV(NR, U, NS, PRO, EPI, NSP, PIPE)                                                                            \
    OVL_GEMV_F(NR, U, NS, PRO, EPI, NSP, PIPE, 0) OVL_GEMV_F(NR, U, NS, PRO, EPI, NSP, PIPE, 1)                                 \
    OVL_GEMV_F(NR, U, NS, PRO, EPI, NSP, PIPE, 2) OVL_GEMV_F(NR, U, NS, PRO, EPI, NSP, PIPE, 3)
#define OVL_ATTN_F(D, WAVES, KVH, F)                                                                                        \
    extern "C" __global__ __launch_bounds__(WAVES * 64) void thk_ovl_attn_##D##_##WAVES##_##KVH##_f##F(const AttnArgs a) {  \
        attn_body<D, WAVES, (KVH) != 0, OVL_QUEUE | F>(a, blockIdx.x);                                                       \
    }
#define OVL_ATTN(D, WAVES, KVH) OVL_ATTN_F(D, WAVES, KVH, 0) OVL_ATTN_F(D, WAVES, KVH, 1) OVL_ATTN_F(D, WAVES, KVH, 2) OVL_ATTN_F(D, WAVES, KVH, 3)

// PRO: 0 copy, 1 RMSNorm, 2 attention combine, 3 RMSNorm of the embedding row; EPI: 1 residual, 2 RoPE + K/V append, 3 SwiGLU, 4 lm-head
// LLaMA-7B widths (4096 / 11008 columns): thk_ctx.cpp auto_geometry's variants + the neighbours worth sweeping
OVL_GEMV(2, 8, 8, 1, 2, 0, 1)   OVL_GEMV(2, 8, 8, 3, 2, 0, 1)       // qkv (pipelined row pair), with the embedding fold
OVL_GEMV(1, 8, 8, 2, 1, 4, 1)   OVL_GEMV(1, 8, 8, 2, 1, 2, 1)  OVL_GEMV(1, 8, 8, 2, 1, 8, 1)      // wo, 4 / 2 / 8 attention splits
OVL_GEMV(2, 8, 8, 1, 3, 0, 1)                                       // w1 | w3
OVL_GEMV(1, 22, 22, 0, 1, 0, 0) OVL_GEMV(1, 22, 22, 0, 1, 0, 1)     // w2
OVL_GEMV(1, 8, 8, 1, 4, 0, 1)   OVL_GEMV(2, 8, 8, 1, 4, 0, 0)       // lm-head
// LLaMA-13B widths (5120 / 13824 columns)
OVL_GEMV(2, 10, 10, 1, 2, 0, 0) OVL_GEMV(2, 10, 10, 3, 2, 0, 0)     // qkv
OVL_GEMV(1, 10, 10, 2, 1, 4, 1) OVL_GEMV(1, 10, 10, 2, 1, 2, 1) OVL_GEMV(1, 10, 10, 2, 1, 8, 1)   // wo
OVL_GEMV(2, 10, 10, 1, 3, 0, 1)                                     // w1 | w3
OVL_GEMV(1, 27, 27, 0, 1, 0, 1) OVL_GEMV(1, 27, 27, 0, 1, 0, 0)     // w2
OVL_GEMV(2, 5, 10, 1, 4, 0, 0)  OVL_GEMV(1, 10, 10, 1, 4, 0, 1)     // lm-head

OVL_ATTN(128, 8, 0) OVL_ATTN(128, 4, 0) OVL_ATTN(128, 8, 1) OVL_ATTN(128, 4, 1)

extern "C" __global__ __launch_bounds__(256) void thk_ovl_finish_token_f0(const FinishArgs a) { finish_token_body<OVL_QUEUE>(a); }
extern "C" __global__ __launch_bounds__(256) void thk_ovl_finish_token_f1(const FinishArgs a) { finish_token_body<OVL_QUEUE | OVL_WAIT>(a); }

// ---- ordering against the HIP stream (thk_ovl.cpp): words[0] = ticket written by a HIP kernel of the ctx stream, words[32] =
// batches this queue has completed, words[64] = error word.
struct OvlGateArgs { unsigned* words; };
// first packet of a batch (barrier bit set): everything the ctx stream had enqueued before the batch has finished once the
// ticket reaches (completed batches + 1)
extern "C" __global__ __launch_bounds__(64) void thk_ovl_batch_begin(const OvlGateArgs a) {
    const unsigned want = __hip_atomic_load(a.words + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1u;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((int)(__hip_atomic_load(a.words, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
        __builtin_amdgcn_s_sleep(32);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 3000000000ull) {          // 30 s: the stream never got there
            if (threadIdx.x == 0) __hip_atomic_store(a.words + 64, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
}
// last packet of a batch (barrier bit set): one more batch done
extern "C" __global__ __launch_bounds__(64) void thk_ovl_batch_end(const OvlGateArgs a) {
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_load(a.words + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.words + 32, done + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
