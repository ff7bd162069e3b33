// engine_probe.hip — does a persistent loader/consumer "engine" keep HBM streaming across DEPENDENT mat-vec ops?
//
// Chain per layer:  y0[12288] = W0 x      (100.7 MB)      x' = y0[0:4096]
//                   y1[22016] = W1 x'     (180.4 MB)      next x = y1[0:4096]
// (a) baseline: one launch per op in a hipGraph (what libthk does today);
// (b) engine: ONE launch, one workgroup per CU.  Wave 0 (loader) streams this CU's rows of every op, in op order,
//     through a ring of 16 KiB LDS slots with global_load_lds (never waits for activations, only for ring space);
//     waves 1-3 (consumers) take the slots round-robin, dot them with the activation vector held in registers and
//     publish results with agent-scope stores; an op's consumers first wait for the previous op's arrival counter.
//     While consumers wait at an op boundary the loader keeps filling the ring (96 KB ~ 4 us of stream per CU).
// Every wait is bounded (error flag) so a protocol bug cannot hang the box.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 engine_probe.hip -o engine_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int C = 4096, NCU = 256;
constexpr int SLOT = 16384;                            // 2 rows of 8 KiB per slot
#ifndef NLOAD
#define NLOAD 1
#endif
constexpr int NCONS = 4 - NLOAD;
constexpr int NSLOT = NLOAD == 1 ? 6 : 8;              // 96 / 128 KiB ring (a multiple of NLOAD and NCONS)
constexpr unsigned SPIN_LIMIT = 1u << 20;

template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
    auto rl = [&](int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
    return (rl(0) + rl(16)) + (rl(32) + rl(48));
}
__device__ __forceinline__ f4 load_f4_sc1(const float* p) {
    const unsigned long long a = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return f4{__builtin_bit_cast(float, (unsigned)a), __builtin_bit_cast(float, (unsigned)(a >> 32)), __builtin_bit_cast(float, (unsigned)b), __builtin_bit_cast(float, (unsigned)(b >> 32))};
}
// dot of one 8 KiB weight row (8 pieces of 1 KiB; lane l owns columns u*512 + 8 l .. +7) with x in registers
__device__ __forceinline__ float row_dot(const h8 (&w)[8], const float (&x)[64]) {
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf((float)w[u][e], x[u * 8 + e], acc);
    return wave_sum(acc);
}

// ---------------------------------------------------------------- baseline: one launch per op (2 rows per wave step, 16 loads in flight)
__global__ __launch_bounds__(256) void op_kernel(const uint16_t* __restrict__ W, int R, const float* __restrict__ x, float* __restrict__ y) {
    const int lane = threadIdx.x & 63, wg = blockIdx.x * 4 + (threadIdx.x >> 6), tw = gridDim.x * 4;
    float xr[64];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const f4 a = *reinterpret_cast<const f4*>(x + u * 512 + lane * 8), b = *reinterpret_cast<const f4*>(x + u * 512 + lane * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { xr[u * 8 + e] = a[e]; xr[u * 8 + 4 + e] = b[e]; }
    }
    const int ng = R / 2;
    h8 w0[8], w1[8];
    auto load = [&](int g) {
        const h8* r0 = reinterpret_cast<const h8*>(W + (size_t)(2 * g) * C) + lane;
#pragma unroll
        for (int u = 0; u < 8; ++u) { w0[u] = __builtin_nontemporal_load(r0 + u * 64); w1[u] = __builtin_nontemporal_load(r0 + 512 + u * 64); }
    };
    if (wg < ng) load(wg);
    for (int g = wg; g < ng; g += tw) {
        h8 c0[8], c1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { c0[u] = w0[u]; c1[u] = w1[u]; }
        if (g + tw < ng) load(g + tw);
        const float a0 = row_dot(c0, xr), a1 = row_dot(c1, xr);
        if (lane == 0) { y[2 * g] = a0; y[2 * g + 1] = a1; }
    }
}

// ---------------------------------------------------------------- engine
struct Op { const uint16_t* W; int slots_per_cu; const float* x; float* y; };   // rows of CU c: [c*2*spc, (c+1)*2*spc)
struct Sync { unsigned done[64 * 32]; unsigned flag[64 * 16 * 32]; unsigned err[32]; };   // per op: arrival counter (own line) + 16 replicated flags

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 2 /* nt */);
}
__device__ __forceinline__ unsigned lds_ld(volatile unsigned* p) { return *p; }

__global__ __launch_bounds__(256, 1) void engine_kernel(const Op* __restrict__ ops, int n_ops, Sync* sy, int dbg) {
    extern __shared__ __attribute__((aligned(1024))) char ring[];          // NSLOT x 16 KiB
    __shared__ volatile unsigned landed[2];                                  // per loader: its slots whose bytes are in LDS (count, in order)
    __shared__ volatile unsigned released[NCONS];                            // slots each consumer has finished (count)
    __shared__ volatile unsigned abort_flag;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) { landed[0] = 0; landed[1] = 0; abort_flag = 0; for (int j = 0; j < NCONS; ++j) released[j] = 0; }
    __syncthreads();
    const int cu = blockIdx.x;

    if (wave < NLOAD) {                                 // ------------------------------------------------ loaders: slot k belongs to loader k % NLOAD
        unsigned k = 0, issued = 0;                     // global slot index of this CU across all ops; slots this loader issued
        for (int p = 0; p < n_ops; ++p) {
            const Op op = ops[p];
            const char* base = reinterpret_cast<const char*>(op.W) + (size_t)cu * op.slots_per_cu * SLOT + lane * 16;
            for (int s = 0; s < op.slots_per_cu; ++s, ++k) {
                if ((int)(k % NLOAD) != wave) continue;
                if (k >= NSLOT) {                       // ring position k % NSLOT was slot k - NSLOT, owned by consumer (k % NCONS): wait for its release
                    const unsigned need = (k - NSLOT) / NCONS + 1;
                    unsigned spins = 0;
                    if (lds_ld(&released[k % NCONS]) < need) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); landed[wave] = issued; }   // blocked anyway: publish all that was issued
                    while (lds_ld(&released[k % NCONS]) < need) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > SPIN_LIMIT || lds_ld(&abort_flag)) { abort_flag = 1; sy->err[0] = 1; return; }
                    }
                }
                char* dst = ring + (k % NSLOT) * SLOT;
                const char* src = base + (size_t)s * SLOT;
#pragma unroll
                for (int i = 0; i < 16; ++i) glds16(src + i * 1024, dst + i * 1024);
                ++issued;
                asm volatile("s_waitcnt vmcnt(47)" ::: "memory");           // <= 47 outstanding => all but my last three slots have landed
                if (issued >= 3 && lds_ld(&landed[wave]) < issued - 3 + 1) landed[wave] = issued - 2;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        landed[wave] = issued;
        return;
    }

    // ---------------------------------------------------------------------------------------------------- consumers
    const int cj = wave - NLOAD;                        // 0..NCONS-1
    unsigned k0 = 0;                                    // global slot index of the op's first slot
    unsigned mine = 0;                                  // slots this consumer has released
    for (int p = 0; p < n_ops; ++p) {
        const Op op = ops[p];
        if (p > 0 && !(dbg & 1)) {                      // wait until every consumer of every CU has published op p-1
            unsigned ok = 1;
            if (lane == 0) {
                const unsigned* f = &sy->flag[((p - 1) * 16 + (cu & 15)) * 32];
                unsigned spins = 0;
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > SPIN_LIMIT || lds_ld(&abort_flag)) { ok = 0; break; }
                }
            }
            ok = __builtin_amdgcn_readfirstlane(ok);
            if (!ok) { abort_flag = 1; sy->err[0] = 2; return; }
        }
        float xr[64];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f4 a = load_f4_sc1(op.x + u * 512 + lane * 8), b = load_f4_sc1(op.x + u * 512 + lane * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { xr[u * 8 + e] = a[e]; xr[u * 8 + 4 + e] = b[e]; }
        }
        const unsigned kend = k0 + op.slots_per_cu;
        // my slots: k in [k0, kend) with k % NCONS == cj
        unsigned k = k0 + ((cj + NCONS - k0 % NCONS) % NCONS);
        for (; k < kend; k += NCONS) {
            unsigned spins = 0;
            while (lds_ld(&landed[k % NLOAD]) <= k / NLOAD) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT || lds_ld(&abort_flag)) { abort_flag = 1; sy->err[0] = 3; return; }
            }
            asm volatile("" ::: "memory");
            const char* sl = ring + (k % NSLOT) * SLOT + lane * 16;
            h8 w0[8], w1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { w0[u] = *reinterpret_cast<const h8*>(sl + u * 1024); w1[u] = *reinterpret_cast<const h8*>(sl + 8192 + u * 1024); }
            float a0 = 0.f, a1 = 0.f;
            if (!(dbg & 2)) { a0 = row_dot(w0, xr); a1 = row_dot(w1, xr); } else { a0 = (float)w0[0][0]; a1 = (float)w1[7][7]; }
            ++mine;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the slot's ds_reads have retired before it is handed back
            if (lane == 0) {
                released[cj] = mine;                    // the ds_reads above have returned (their data was consumed)
                const size_t row = ((size_t)cu * op.slots_per_cu + (k - k0)) * 2;
                __hip_atomic_store(op.y + row, a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(op.y + row + 1, a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // publish: drain my stores, bump the op's arrival counter; the last arriver raises the 16 replicated flags
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const unsigned t = __hip_atomic_fetch_add(&sy->done[p * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == NCU * NCONS - 1)
                for (int i = 0; i < 16; ++i) __hip_atomic_store(&sy->flag[(p * 16 + i) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        k0 = kend;
    }
}

int main(int argc, char** argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 16;
    if (2 * layers > 64) { printf("at most 32 layers\n"); return 1; }
    const int R0 = 12288, R1 = 22016;
    const size_t b0 = (size_t)R0 * C * 2, b1 = (size_t)R1 * C * 2;
    const int NBUF = 8;                                 // rotate over 8 x 281 MB so nothing is cache-resident
    uint16_t* pool; CHECK(hipMalloc(&pool, (b0 + b1) * NBUF));
    {   // weights: small pseudo-random f16 values (|w| < 2^-6) so that the chain neither explodes nor vanishes quickly
        std::vector<uint16_t> h((b0 + b1) / 2);
        uint32_t s = 12345;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x1c00 + ((s >> 16) & 0x3ff) + ((s >> 31) << 15)); }   // +-[2^-8, 2^-7)
        for (int i = 0; i < NBUF; ++i) CHECK(hipMemcpy(pool + (size_t)i * (b0 + b1) / 2, h.data(), b0 + b1, hipMemcpyHostToDevice));
    }
    float *x0, *ya, *yb; CHECK(hipMalloc(&x0, C * 4)); CHECK(hipMalloc(&ya, (size_t)R1 * 4)); CHECK(hipMalloc(&yb, (size_t)R1 * 4));
    std::vector<float> hx(C); for (int i = 0; i < C; ++i) hx[i] = 0.5f + 0.001f * (i % 97);
    CHECK(hipMemcpy(x0, hx.data(), C * 4, hipMemcpyHostToDevice));
    Sync* sy; CHECK(hipMalloc(&sy, sizeof(Sync)));
    std::vector<Op> hop;
    for (int l = 0; l < layers; ++l) {
        const uint16_t* base = pool + (size_t)(l % NBUF) * (b0 + b1) / 2;
        hop.push_back({base, R0 / 2 / NCU, l == 0 ? x0 : yb, ya});          // op0 reads previous y1 (yb), writes ya
        hop.push_back({base + b0 / 2, R1 / 2 / NCU, ya, yb});              // op1 reads ya[0:4096], writes yb
    }
    Op* dop; CHECK(hipMalloc(&dop, hop.size() * sizeof(Op))); CHECK(hipMemcpy(dop, hop.data(), hop.size() * sizeof(Op), hipMemcpyHostToDevice));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<float> ref(R1), got(R1);
    auto timeit = [&](auto fn, const char* name, std::vector<float>& out) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipMemsetAsync(sy, 0, sizeof(Sync), st));
            CHECK(hipEventRecord(e0, st)); fn(); CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        unsigned err = 0; CHECK(hipMemcpy(&err, &sy->err[0], 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(out.data(), yb, (size_t)R1 * 4, hipMemcpyDeviceToHost));
        const double mb = (double)(b0 + b1) * layers / 1e6;
        printf("%-30s %8.1f us total, %6.2f us/layer, %6.3f TB/s  err=%u  y[0]=%g\n", name, best * 1e3, best * 1e3 / layers, mb / (best * 1e3), err, out[0]);
    };
    hipGraph_t graph; hipGraphExec_t exec;
    const int bgrid = argc > 2 ? atoi(argv[2]) : 1024;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (size_t i = 0; i < hop.size(); ++i) hipLaunchKernelGGL(op_kernel, dim3(bgrid), dim3(256), 0, st, hop[i].W, hop[i].slots_per_cu * 2 * NCU, hop[i].x, hop[i].y);
    CHECK(hipStreamEndCapture(st, &graph)); CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    timeit([&] { CHECK(hipGraphLaunch(exec, st)); }, "separate launches (hipGraph)", ref);
    CHECK(hipFuncSetAttribute((const void*)engine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT));
    timeit([&] { hipLaunchKernelGGL(engine_kernel, dim3(NCU), dim3(256), NSLOT * SLOT, st, dop, (int)hop.size(), sy, 0); }, "engine (1 launch)", got);
    std::vector<float> tmp(R1);
    timeit([&] { hipLaunchKernelGGL(engine_kernel, dim3(NCU), dim3(256), NSLOT * SLOT, st, dop, (int)hop.size(), sy, 1); }, "engine, no hand-off wait", tmp);
    timeit([&] { hipLaunchKernelGGL(engine_kernel, dim3(NCU), dim3(256), NSLOT * SLOT, st, dop, (int)hop.size(), sy, 2); }, "engine, no compute", tmp);
    timeit([&] { hipLaunchKernelGGL(engine_kernel, dim3(NCU), dim3(256), NSLOT * SLOT, st, dop, (int)hop.size(), sy, 3); }, "engine, neither", tmp);
    size_t bad = 0; for (int i = 0; i < R1; ++i) if (ref[i] != got[i]) ++bad;
    printf("engine vs launches: %zu of %d outputs differ\n", bad, R1);
    return 0;
}
