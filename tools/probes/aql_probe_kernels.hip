// aql_probe_kernels.hip — device side of tools/probes/aql_overlap_probe.cpp (round 3).
// A chain of dependent weight-streaming mat-vecs (4096 columns, a row pair per wave step, the decode path's batch loop), written
// so that it can run two ways:
//   wait_mode 0  the usual way: the AQL packet carries the barrier bit, the vector is read with plain loads
//   wait_mode 1  the packet carries NO barrier bit (the command processor may start it while its predecessor still runs): every
//                wave requests its first weight batch, then the workgroup waits until the predecessor's arrival counter reaches
//                its grid size (agent-scope loads, bounded), reads the vector with agent-scope loads, and at its own end drains
//                its stores and bumps its own counter.
// No gridDim / blockDim builtins: the host does not fill the hidden kernel arguments.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 --genco aql_probe_kernels.hip -o aql_probe_kernels.hsaco
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int C = 4096, NS = 8, THREADS = 256, WAVES = 4;

template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
    auto rl = [&](int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
    return (rl(0) + rl(16)) + (rl(32) + rl(48));
}
// one 16-byte agent-coherent (sc1) load: raw buffer form so that it stays ONE instruction
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 ld_f4_agent(const float* base, unsigned byte_off) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 1 << 20, 0x00020000);
    const u4v t = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16 /* sc1 */);
    return __builtin_bit_cast(f4, t);     // whole-vector cast: __builtin_bit_cast of ONE element of an ext_vector (t.y ...) reads element 0 with this
                                          // toolchain, which made the first version of this probe fetch 4 of every 16 bytes (results of 27 Sep, see profiles/)
}

struct ProbeArgs {
    const uint16_t* W;            // [2 * n_groups][C] f16
    int n_groups, n_blocks;
    const float* x_in;            // [C]
    float* y_out;                 // [2 * n_groups]
    const unsigned* prev_done;    // wait_mode 1: the predecessor's arrival counters (n_shards words, 32 words = 128 bytes apart) ...
    unsigned prev_target;         // ... and the total they reach when it is done (counters only grow)
    unsigned* my_done;            // wait_mode 1: this launch's arrival counters
    int wait_mode;
    unsigned* err;
    int n_shards;                 // 1 .. 64 (power of two)
    int poll_sleep;               // s_sleep argument between polls
    int first_sleeps;             // s_sleep(32) repetitions before the first poll
};

extern "C" __global__ __launch_bounds__(THREADS) void gemv_dep(const ProbeArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[NS * 512];
    __shared__ int ok;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int total_waves = a.n_blocks * WAVES;
    int g = blockIdx.x * WAVES + wave;
    const bool has = g < a.n_groups;
    const int g0 = has ? g : a.n_groups - 1;
    h8 w[2][NS];
    {   // first batch: independent of the predecessor
        const h8* r0 = reinterpret_cast<const h8*>(a.W + (size_t)(2 * g0) * C);
        const h8* r1 = reinterpret_cast<const h8*>(a.W + (size_t)(2 * g0 + 1) * C);
#pragma unroll
        for (int u = 0; u < NS; ++u) { w[0][u] = __builtin_nontemporal_load(r0 + u * 64 + lane); w[1][u] = __builtin_nontemporal_load(r1 + u * 64 + lane); }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (a.wait_mode) {
        if (wave == 0) {                                          // lanes < n_shards each watch one shard; the wave adds them up
            int good = 1;
            if (a.prev_done) {
                for (int k = 0; k < a.first_sleeps; ++k) __builtin_amdgcn_s_sleep(32);
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    unsigned v = lane < a.n_shards ? __hip_atomic_load(a.prev_done + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                    if (v >= a.prev_target) break;
                    __builtin_amdgcn_s_sleep(8);
                    for (int k = 1; k < a.poll_sleep; ++k) __builtin_amdgcn_s_sleep(8);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) { good = 0; if (lane == 0) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
            }
            if (lane == 0) ok = good;
        }
        __syncthreads();
        if (!ok) return;
    }
    for (int i = threadIdx.x; i < C / 4; i += THREADS) {       // stage the vector (lo/hi split layout of the decode kernels)
        const f4 v = a.wait_mode ? ld_f4_agent(a.x_in, (unsigned)i * 16u) : *reinterpret_cast<const f4*>(a.x_in + i * 4);
        const int e = i * 4, gg = e >> 3, j = e & 7;
        *reinterpret_cast<f4*>(xs + (j < 4 ? 0 : NS * 256) + (gg << 2)) = v;
    }
    __syncthreads();
    const f4* xlo = reinterpret_cast<const f4*>(xs);
    const f4* xhi = reinterpret_cast<const f4*>(xs + NS * 256);
    bool first = true;
    for (; g < a.n_groups; g += total_waves) {
        if (!first) {
            const h8* r0 = reinterpret_cast<const h8*>(a.W + (size_t)(2 * g) * C);
            const h8* r1 = reinterpret_cast<const h8*>(a.W + (size_t)(2 * g + 1) * C);
#pragma unroll
            for (int u = 0; u < NS; ++u) { w[0][u] = __builtin_nontemporal_load(r0 + u * 64 + lane); w[1][u] = __builtin_nontemporal_load(r1 + u * 64 + lane); }
            __builtin_amdgcn_sched_barrier(0);
        }
        first = false;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const f4 xl = xlo[u * 64 + lane], xh = xhi[u * 64 + lane];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float acc = r == 0 ? a0 : a1;
                const h8 ww = w[r][u];
                acc = fmaf((float)ww[0], xl.x, acc); acc = fmaf((float)ww[1], xl.y, acc); acc = fmaf((float)ww[2], xl.z, acc); acc = fmaf((float)ww[3], xl.w, acc);
                acc = fmaf((float)ww[4], xh.x, acc); acc = fmaf((float)ww[5], xh.y, acc); acc = fmaf((float)ww[6], xh.z, acc); acc = fmaf((float)ww[7], xh.w, acc);
                if (r == 0) a0 = acc; else a1 = acc;
            }
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1);
        if (lane == 0) {
            if (a.wait_mode) {
                __hip_atomic_store(a.y_out + 2 * g, a0 * 1e-3f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.y_out + 2 * g + 1, a1 * 1e-3f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else { a.y_out[2 * g] = a0 * 1e-3f; a.y_out[2 * g + 1] = a1 * 1e-3f; }
        }
    }
    if (a.wait_mode) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave drains its write-through stores ...
        __syncthreads();
        if (threadIdx.x == 0)                                   // ... then the workgroup arrives on its shard (no return value: fire and forget)
            (void)__hip_atomic_fetch_add(a.my_done + (blockIdx.x & (a.n_shards - 1)) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

extern "C" __global__ void zero_words(unsigned* p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
}
