// persist_probe.hip — experiment: does a persistent launch with a grid barrier between ops keep
// HBM busy across op boundaries if every wave prefetches its first two weight batches of the NEXT
// op before arriving at the barrier?  Compared against the same streaming loop as separate launches.
//   phases per "layer": P0 = 12288 rows x 4096 (100.7 MB), P1 = 22016 rows x 4096 (180.4 MB)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 persist_probe.hip -o persist_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int C = 4096, NS = 8, NR = 2;          // 8 slots per row, 2 rows per batch => 16 loads per lane
constexpr int THREADS = 512, WAVES = THREADS / 64;

struct Batch { h8 w[NR][NS]; };

template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
    auto rl = [&](int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
    return (rl(0) + rl(16)) + (rl(32) + rl(48));
}
__device__ __forceinline__ void load_batch(Batch& b, const uint16_t* W, int g, int lane) {
    const h8* r0 = reinterpret_cast<const h8*>(W + (size_t)(2 * g) * C);
    const h8* r1 = reinterpret_cast<const h8*>(W + (size_t)(2 * g + 1) * C);
#pragma unroll
    for (int u = 0; u < NS; ++u) { b.w[0][u] = __builtin_nontemporal_load(r0 + u * 64 + lane); b.w[1][u] = __builtin_nontemporal_load(r1 + u * 64 + lane); }
}
__device__ __forceinline__ void compute_batch(const Batch& b, const f4* xlo, const f4* xhi, int lane, float& a0, float& a1) {
    a0 = 0.f; a1 = 0.f;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const f4 xl = xlo[u * 64 + lane], xh = xhi[u * 64 + lane];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float acc = r == 0 ? a0 : a1;
            const h8 w = b.w[r][u];
            acc = fmaf((float)w[0], xl.x, acc); acc = fmaf((float)w[1], xl.y, acc); acc = fmaf((float)w[2], xl.z, acc); acc = fmaf((float)w[3], xl.w, acc);
            acc = fmaf((float)w[4], xh.x, acc); acc = fmaf((float)w[5], xh.y, acc); acc = fmaf((float)w[6], xh.z, acc); acc = fmaf((float)w[7], xh.w, acc);
            if (r == 0) a0 = acc; else a1 = acc;
        }
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1);
}
__device__ __forceinline__ void stage_x(float* xs, const float* x) {    // plain copy into the split lo/hi layout
    for (int i = threadIdx.x; i < C / 4; i += THREADS) {
        const f4 v = *reinterpret_cast<const f4*>(x + i * 4);
        const int e = i * 4, g = e >> 3, j = e & 7;
        *reinterpret_cast<f4*>(xs + (j < 4 ? 0 : NS * 256) + (g << 2)) = v;
    }
    __syncthreads();
}

// ---------------------------------------------------------------- baseline: one launch per phase
__global__ __launch_bounds__(THREADS) void phase_kernel(const uint16_t* W, int n_groups, const float* x, float* y) {
    __shared__ __attribute__((aligned(16))) float xs[NS * 512];
    const int lane = threadIdx.x & 63, wg = blockIdx.x * WAVES + (threadIdx.x >> 6), tw = gridDim.x * WAVES;
    Batch b;
    if (wg < n_groups) load_batch(b, W, wg, lane);
    stage_x(xs, x);
    const f4* xlo = reinterpret_cast<const f4*>(xs); const f4* xhi = reinterpret_cast<const f4*>(xs + NS * 256);
    for (int g = wg; g < n_groups; g += tw) {
        float a0, a1;
        compute_batch(b, xlo, xhi, lane, a0, a1);
        const int gn = g + tw;
        if (gn < n_groups) load_batch(b, W, gn, lane);
        if (lane == 0) { y[2 * g] = a0; y[2 * g + 1] = a1; }
    }
}

// ---------------------------------------------------------------- grid barrier (two-level, bounded)
struct Bar { unsigned group[8 * 32]; unsigned top[32]; unsigned gen[8 * 32]; unsigned err[32]; };
__device__ __forceinline__ void grid_barrier(Bar* bar, unsigned epoch /* 1,2,3.. */, int nblocks) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int grp = blockIdx.x & 7, per = nblocks >> 3;
        const unsigned t = __hip_atomic_fetch_add(&bar->group[grp * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == epoch * per - 1) {                                   // last of my group this epoch
            const unsigned tt = __hip_atomic_fetch_add(&bar->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tt == epoch * 8 - 1)
                for (int k = 0; k < 8; ++k) __hip_atomic_store(&bar->gen[k * 32], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned spins = 0;
        while (__hip_atomic_load(&bar->gen[grp * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) { __hip_atomic_store(&bar->err[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------- persistent: all phases of all layers in one launch
struct Phase { const uint16_t* W; int n_groups; float* y; };
template <bool PREFETCH>
__global__ __launch_bounds__(THREADS) void persist_kernel(const Phase* phases, int n_phases, const float* x, Bar* bar) {
    __shared__ __attribute__((aligned(16))) float xs[NS * 512];
    const int lane = threadIdx.x & 63, wg = blockIdx.x * WAVES + (threadIdx.x >> 6), tw = gridDim.x * WAVES;
    const f4* xlo = reinterpret_cast<const f4*>(xs); const f4* xhi = reinterpret_cast<const f4*>(xs + NS * 256);
    Batch A, B;          // A = next batch to compute, B = the one after (both for the CURRENT phase on entry)
    {
        const Phase p = phases[0];
        if (wg < p.n_groups) load_batch(A, p.W, wg, lane);
        if (wg + tw < p.n_groups) load_batch(B, p.W, wg + tw, lane);
    }
    bool a_first = true;
    for (int ph = 0; ph < n_phases; ++ph) {
        const Phase p = phases[ph];
        const Phase pn = phases[ph + 1 < n_phases ? ph + 1 : ph];
        const bool has_next = ph + 1 < n_phases;
        stage_x(xs, x);
        // batches of this wave: g_k = wg + k*tw.  After computing batch k its registers take batch k+2 of this
        // phase, or (PREFETCH) batch (k+2-n_k) of the next phase when this phase has run out.
        int nk = p.n_groups > wg ? (p.n_groups - wg + tw - 1) / tw : 0;
        int next_issued = 0;                                   // next-phase batches already in flight (0..2)
        for (int k = 0; k < nk; ++k) {
            const int g = wg + k * tw;
            float a0, a1;
            const bool useA = a_first ? ((k & 1) == 0) : ((k & 1) == 1);
            if (useA) compute_batch(A, xlo, xhi, lane, a0, a1); else compute_batch(B, xlo, xhi, lane, a0, a1);
            const int g2 = g + 2 * tw;
            if (g2 < p.n_groups) { if (useA) load_batch(A, p.W, g2, lane); else load_batch(B, p.W, g2, lane); }
            else if (PREFETCH && has_next) {
                const int gn = wg + next_issued * tw;
                if (gn < pn.n_groups) { if (useA) load_batch(A, pn.W, gn, lane); else load_batch(B, pn.W, gn, lane); }
                ++next_issued;
            }
            if (lane == 0) { __hip_atomic_store(p.y + 2 * g, a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(p.y + 2 * g + 1, a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        // which register set holds the next phase's batch 0?  it replaced batch nk-2 (if nk>=2) else the loads happen below
        bool next_a_first;
        if (nk >= 2) { const int k0 = nk - 2; const bool k0_useA = a_first ? ((k0 & 1) == 0) : ((k0 & 1) == 1); next_a_first = k0_useA; }
        else if (nk == 1) { const bool k0_useA = a_first; next_a_first = k0_useA; /* batch0 of next went where batch 0 was */ }
        else next_a_first = true;
        grid_barrier(bar, (unsigned)ph + 1, gridDim.x);
        if (has_next) {
            // anything not prefetched is loaded now
            int have = PREFETCH ? next_issued : 0;
            if (nk == 0) have = 0;
            for (int q = have; q < 2; ++q) {
                const int gn = wg + q * tw;
                const bool intoA = (q == 0) ? next_a_first : !next_a_first;
                if (gn < pn.n_groups) { if (intoA) load_batch(A, pn.W, gn, lane); else load_batch(B, pn.W, gn, lane); }
            }
        }
        a_first = next_a_first;
    }
}

int main(int argc, char** argv) {
    const int layers = 32, bpc = argc > 1 ? atoi(argv[1]) : 1;
    int dev = 0; hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, dev));
    const int ncu = prop.multiProcessorCount;
    const int R0 = 12288, R1 = 22016;
    const size_t b0 = (size_t)R0 * C * 2, b1 = (size_t)R1 * C * 2;
    uint16_t* pool; const int NBUF = 8;                    // rotate over 8 x 281 MB = 2.2 GB so nothing is cache-resident
    CHECK(hipMalloc(&pool, (b0 + b1) * NBUF));
    CHECK(hipMemset(pool, 0x11, (b0 + b1) * NBUF));
    float *x, *y; CHECK(hipMalloc(&x, C * 4)); CHECK(hipMalloc(&y, (size_t)(R0 + R1) * 4 * 2));
    std::vector<float> hx(C, 0.01f); CHECK(hipMemcpy(x, hx.data(), C * 4, hipMemcpyHostToDevice));
    Bar* bar; CHECK(hipMalloc(&bar, sizeof(Bar)));
    std::vector<Phase> hp;
    for (int l = 0; l < layers; ++l) {
        uint16_t* base = pool + (size_t)(l % NBUF) * (b0 + b1) / 2;
        hp.push_back({base, R0 / 2, y}); hp.push_back({base + b0 / 2, R1 / 2, y + R0});
    }
    Phase* dp; CHECK(hipMalloc(&dp, hp.size() * sizeof(Phase))); CHECK(hipMemcpy(dp, hp.data(), hp.size() * sizeof(Phase), hipMemcpyHostToDevice));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = ncu * bpc;
    auto timeit = [&](auto fn, const char* name) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipMemsetAsync(bar, 0, sizeof(Bar), st));
            CHECK(hipEventRecord(e0, st)); fn(); CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        unsigned err = 0; CHECK(hipMemcpy(&err, &bar->err[0], 4, hipMemcpyDeviceToHost));
        const double mb = (double)(b0 + b1) * layers / 1e6;
        printf("%-34s grid %4d: %8.1f us total, %6.2f us/layer, %6.3f TB/s  err=%u\n", name, grid, best * 1e3, best * 1e3 / layers, mb / (best * 1e3) / 1e6 * 1e6 / 1e6, err);
    };
    // baseline as a captured graph of 64 launches (what libthk does today)
    hipGraph_t graph; hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (auto& p : hp) hipLaunchKernelGGL(phase_kernel, dim3(grid), dim3(THREADS), 0, st, p.W, p.n_groups, x, p.y);
    CHECK(hipStreamEndCapture(st, &graph)); CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    timeit([&] { CHECK(hipGraphLaunch(exec, st)); }, "separate launches (hipGraph)");
    timeit([&] { hipLaunchKernelGGL(persist_kernel<false>, dim3(grid), dim3(THREADS), 0, st, dp, (int)hp.size(), x, bar); }, "persistent, barrier, no prefetch");
    timeit([&] { hipLaunchKernelGGL(persist_kernel<true>, dim3(grid), dim3(THREADS), 0, st, dp, (int)hp.size(), x, bar); }, "persistent, barrier + 2-batch prefetch");
    return 0;
}
