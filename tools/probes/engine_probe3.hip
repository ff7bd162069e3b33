// engine_probe3.hip — round-2 redo of the persistent loader/consumer engine probe, following the guide's recipe
// (MI355X_MICROARCH.md price list rows ldsdma-fill, prefetch-credit, allgather, engine-vs-launches).
//
// Why round 1's probe measured half the documented fill rate: its LDS control words were `volatile` objects read
// through generic pointers, which hipcc compiled to `flat_load_dword ... sc0 sc1` + `s_waitcnt vmcnt(0)` — every ring
// poll drained the loader's LDS-DMA queue.  Here every LDS control access is a hand-written ds_read/ds_write (lgkmcnt
// only) and the loader's only vmcnt waits are the counted ones below.
//
// Chain per layer (all C = 4096, so the op sizes are a 7B layer's): 12288 rows (qkv), 4096 (wo), 22016 (w1|w3), 11264 (~w2)
// next x = first 4096 outputs.  (a) baseline: one launch per op in a hipGraph.  (b) engine: ONE launch, one workgroup per
// CU, wave 0 = loader (16 KiB fills of 16 x 1 KiB global_load_lds into an NS-slot ring, DEPTH fills in flight), waves 1-3 =
// consumers (slot k -> consumer k % 3), outputs published as 8-byte {value, tag} granules with agent-scope stores,
// gathered by consumer 0 of every CU into LDS (flat sweep, re-read until every tag matches), copied to registers.
// Every wait is bounded; a time-out raises sy->err and aborts the workgroup.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 engine_probe3.hip -o engine_probe3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
constexpr int C = 4096, NCU = 256, SLOT = 16384, NCONS = 3;
constexpr unsigned SPIN_LIMIT = 1u << 21;

template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
    auto rl = [&](int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
    return (rl(0) + rl(16)) + (rl(32) + rl(48));
}
__device__ __forceinline__ float row_dot(const h8 (&w)[8], const float (&x)[64]) {
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf((float)w[u][e], x[u * 8 + e], acc);
    return wave_sum(acc);
}

// ---------------------------------------------------------------- baseline: one launch per op
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
struct Op { const uint16_t* W; int spc; int pad; float* y; };   // spc = 16 KiB slots (row pairs) per CU
struct Sync { unsigned err[64]; };

// LDS control words, addressed by byte offset (ds_* instructions; never through generic pointers)
__device__ __forceinline__ unsigned lds_ld(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_st(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_inc(unsigned addr) { unsigned one = 1; asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(one) : "memory"); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int AUX>
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, AUX);
}

enum { CT_LANDED = 0, CT_REL = 4 /* 3 words */, CT_ABORT = 16, CT_XREADY = 20, CT_XTAKEN = 24, CT_BYTES = 64 };

// dbg bit0: no dependency (x0 for every op), bit1: no FMAs, bit2: consumers release without reading, bit3: interleaved slot->address map
template <int DEPTH, int NS, int AUX>
__global__ __launch_bounds__(256, 1) void engine_kernel(const Op* __restrict__ ops, int n_ops, const float* __restrict__ x0, u64* xg, Sync* sy, int dbg) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];          // [NS x 16 KiB ring][xs 16 KiB][control]
    char* ring = smem;
    float* xs = reinterpret_cast<float*>(smem + NS * SLOT);
    const unsigned ctl = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + NS * SLOT + C * 4);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < CT_BYTES / 4) lds_st(ctl + threadIdx.x * 4, 0u);
    __syncthreads();
    const int cu = blockIdx.x;
    const bool ilv = dbg & 8;

    if (wave == 0) {   // ------------------------------------------------------------------ loader
        unsigned k = 0;
        for (int p = 0; p < n_ops; ++p) {
            const Op op = ops[p];
            const char* base = reinterpret_cast<const char*>(op.W) + lane * 16;
            for (int s = 0; s < op.spc; ++s, ++k) {
                if (k >= (unsigned)NS) {     // ring position k % NS still holds slot k - NS: wait until its consumer released it
                    const unsigned need = (k - NS) / NCONS + 1, addr = ctl + CT_REL + ((k - NS) % NCONS) * 4;
                    if (lds_ld(addr) < need) {
                        wait_vm<0>(); lds_st(ctl + CT_LANDED, k);            // blocked anyway: everything issued has landed by then
                        unsigned spins = 0;
                        while (lds_ld(addr) < need) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > SPIN_LIMIT || lds_ld(ctl + CT_ABORT)) { lds_st(ctl + CT_ABORT, 1u); if (lane == 0) sy->err[0] = 1; return; }
                        }
                    }
                }
                const size_t gs = ilv ? ((size_t)s * NCU + cu) : ((size_t)cu * op.spc + s);
                const char* src = base + gs * SLOT;
                char* dst = ring + (k % NS) * SLOT;
#pragma unroll
                for (int i = 0; i < 16; ++i) glds16<AUX>(src + i * 1024, dst + i * 1024);
                wait_vm<16 * (DEPTH - 1)>();                                  // <= DEPTH-1 fills outstanding
                if (k + 1 >= (unsigned)DEPTH) lds_st(ctl + CT_LANDED, k + 2 - DEPTH);
            }
        }
        wait_vm<0>();
        lds_st(ctl + CT_LANDED, k);
        return;
    }

    // ---------------------------------------------------------------------------------------- consumers
    const int cj = wave - 1;
    unsigned k0 = 0, mine = 0;
    for (int p = 0; p < n_ops; ++p) {
        const Op op = ops[p];
        float xr[64];
        if (p == 0 || (dbg & 1)) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f4 a = *reinterpret_cast<const f4*>(x0 + u * 512 + lane * 8), b = *reinterpret_cast<const f4*>(x0 + u * 512 + lane * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { xr[u * 8 + e] = a[e]; xr[u * 8 + 4 + e] = b[e]; }
            }
        } else {
            if (cj == 0) {    // gatherer: granules of op p-1 (tag p) -> xs
                unsigned spins = 0;
                while (lds_ld(ctl + CT_XTAKEN) < (unsigned)(2 * (p - 1)) && p > 1) {   // the other consumers hold the previous x in registers
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT || lds_ld(ctl + CT_ABORT)) { lds_st(ctl + CT_ABORT, 1u); if (lane == 0) sy->err[0] = 2; return; }
                }
                const u64* g = xg + (size_t)(p & 1) * C;
                for (int ch = 0; ch < C / 1024; ++ch) {
                    unsigned v[16];
                    spins = 0;
                    for (;;) {
                        bool ok = true;
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const u64 t = __hip_atomic_load(g + ch * 1024 + j * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            v[j] = (unsigned)t; ok &= (unsigned)(t >> 32) == (unsigned)p;
                        }
                        if (__all(ok)) break;
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > SPIN_LIMIT || lds_ld(ctl + CT_ABORT)) { lds_st(ctl + CT_ABORT, 1u); if (lane == 0) sy->err[0] = 3; return; }
                    }
#pragma unroll
                    for (int j = 0; j < 16; ++j) xs[ch * 1024 + j * 64 + lane] = __builtin_bit_cast(float, v[j]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                lds_st(ctl + CT_XREADY, (unsigned)p);
            } else {
                unsigned spins = 0;
                while (lds_ld(ctl + CT_XREADY) < (unsigned)p) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT || lds_ld(ctl + CT_ABORT)) { lds_st(ctl + CT_ABORT, 1u); if (lane == 0) sy->err[0] = 4; return; }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f4 a = *reinterpret_cast<const f4*>(xs + u * 512 + lane * 8), b = *reinterpret_cast<const f4*>(xs + u * 512 + lane * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { xr[u * 8 + e] = a[e]; xr[u * 8 + 4 + e] = b[e]; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (cj != 0 && lane == 0) lds_inc(ctl + CT_XTAKEN);
        }
        const unsigned kend = k0 + op.spc;
        u64* gout = xg + (size_t)((p + 1) & 1) * C;
        for (unsigned k = k0 + ((cj + NCONS - k0 % NCONS) % NCONS); k < kend; k += NCONS) {
            unsigned spins = 0;
            while (lds_ld(ctl + CT_LANDED) <= k) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT || lds_ld(ctl + CT_ABORT)) { lds_st(ctl + CT_ABORT, 1u); if (lane == 0) sy->err[0] = 5; return; }
            }
            float a0 = 0.f, a1 = 0.f;
            if (!(dbg & 4)) {
                const char* sl = ring + (k % NS) * SLOT + lane * 16;
                h8 w0[8], w1[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { w0[u] = *reinterpret_cast<const h8*>(sl + u * 1024); w1[u] = *reinterpret_cast<const h8*>(sl + 8192 + u * 1024); }
                if (!(dbg & 2)) { a0 = row_dot(w0, xr); a1 = row_dot(w1, xr); } else { a0 = (float)w0[0][0] + (float)w0[7][7]; a1 = (float)w1[0][0] + (float)w1[7][7]; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            ++mine;
            lds_st(ctl + CT_REL + cj * 4, mine);
            if (lane == 0) {
                const unsigned s = k - k0;
                const size_t row = (ilv ? ((size_t)s * NCU + cu) : ((size_t)cu * op.spc + s)) * 2;
                op.y[row] = a0; op.y[row + 1] = a1;
                if (row < (size_t)C) {
                    __hip_atomic_store(gout + row, ((u64)(unsigned)(p + 1) << 32) | __builtin_bit_cast(unsigned, a0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(gout + row + 1, ((u64)(unsigned)(p + 1) << 32) | __builtin_bit_cast(unsigned, a1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        k0 = kend;
    }
}

struct Variant { const char* name; void (*fn)(const Op*, int, const float*, u64*, Sync*, int); int ns; };

int main(int argc, char** argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 16;
    const int bgrid = argc > 2 ? atoi(argv[2]) : 1024;
    const int RS[4] = {12288, 4096, 22016, 11264};
    size_t bytes_layer = 0, off[4];
    for (int i = 0; i < 4; ++i) { off[i] = bytes_layer; bytes_layer += (size_t)RS[i] * C * 2; }
    const int NBUF = 4;                                  // rotate over 4 x 407 MB so nothing is cache-resident
    char* pool; CHECK(hipMalloc(&pool, bytes_layer * NBUF));
    {
        std::vector<uint16_t> h(bytes_layer / 2);
        uint32_t s = 12345;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x2000 + ((s >> 16) & 0x3ff) + ((s >> 31) << 15)); }   // +-[2^-7, 2^-6)
        for (int i = 0; i < NBUF; ++i) CHECK(hipMemcpy(pool + (size_t)i * bytes_layer, h.data(), bytes_layer, hipMemcpyHostToDevice));
    }
    float *x0, *ya, *yb; CHECK(hipMalloc(&x0, C * 4)); CHECK(hipMalloc(&ya, (size_t)22016 * 4)); CHECK(hipMalloc(&yb, (size_t)22016 * 4));
    std::vector<float> hx(C); for (int i = 0; i < C; ++i) hx[i] = 0.5f + 0.001f * (i % 97);
    CHECK(hipMemcpy(x0, hx.data(), C * 4, hipMemcpyHostToDevice));
    Sync* sy; CHECK(hipMalloc(&sy, sizeof(Sync)));
    u64* xg; CHECK(hipMalloc(&xg, 2 * C * 8));
    std::vector<Op> hop;
    for (int l = 0; l < layers; ++l)
        for (int i = 0; i < 4; ++i) {
            const int p = l * 4 + i;
            hop.push_back({reinterpret_cast<const uint16_t*>(pool + (size_t)(l % NBUF) * bytes_layer + off[i]), RS[i] / 2 / NCU, 0, (p & 1) ? yb : ya});
        }
    const int n_ops = (int)hop.size();
    Op* dop; CHECK(hipMalloc(&dop, hop.size() * sizeof(Op))); CHECK(hipMemcpy(dop, hop.data(), hop.size() * sizeof(Op), hipMemcpyHostToDevice));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<float> ref(22016), got(22016);
    float* ylast = ((n_ops - 1) & 1) ? yb : ya;
    const double mb = (double)bytes_layer * layers / 1e6;
    auto timeit = [&](auto fn, const char* name, std::vector<float>& out) {
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipMemsetAsync(sy, 0, sizeof(Sync), st));
            CHECK(hipMemsetAsync(xg, 0, 2 * C * 8, st));
            CHECK(hipEventRecord(e0, st)); fn(); CHECK(hipEventRecord(e1, st));
            hipError_t e = hipStreamSynchronize(st);
            if (e != hipSuccess) { printf("%s: sync failed: %s\n", name, hipGetErrorString(e)); exit(1); }
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        unsigned err = 0; CHECK(hipMemcpy(&err, &sy->err[0], 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(out.data(), ylast, (size_t)22016 * 4, hipMemcpyDeviceToHost));
        printf("%-44s %8.1f us, %6.2f us/layer, %6.3f TB/s  err=%u  y[0]=%g\n", name, best * 1e3, best * 1e3 / layers, mb / (best * 1e3), err, out[0]);
        fflush(stdout);
    };
    hipGraph_t graph; hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < n_ops; ++p) {
        const float* xin = p == 0 ? x0 : hop[p - 1].y;
        hipLaunchKernelGGL(op_kernel, dim3(bgrid), dim3(256), 0, st, hop[p].W, RS[p % 4], xin, hop[p].y);
    }
    CHECK(hipStreamEndCapture(st, &graph)); CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    timeit([&] { CHECK(hipGraphLaunch(exec, st)); }, "separate launches (hipGraph)", ref);

    const Variant vars[] = {
        {"D2 NS8 nt", engine_kernel<2, 8, 2>, 8}, {"D3 NS8 nt", engine_kernel<3, 8, 2>, 8}, {"D4 NS8 nt", engine_kernel<4, 8, 2>, 8},
        {"D3 NS6 nt", engine_kernel<3, 6, 2>, 6}, {"D3 NS8 default", engine_kernel<3, 8, 0>, 8}, {"D2 NS6 nt", engine_kernel<2, 6, 2>, 6},
    };
    std::vector<float> tmp(22016);
    for (const Variant& v : vars) {
        const size_t lds = (size_t)v.ns * SLOT + C * 4 + CT_BYTES;
        CHECK(hipFuncSetAttribute((const void*)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        char nm[96];
        for (int ilv = 0; ilv < 2; ++ilv) {
            const int b = ilv ? 8 : 0;
            snprintf(nm, sizeof nm, "%s ilv%d loader only", v.name, ilv);
            timeit([&] { hipLaunchKernelGGL(v.fn, dim3(NCU), dim3(256), lds, st, dop, n_ops, x0, xg, sy, 1 | 4 | b); }, nm, tmp);
            snprintf(nm, sizeof nm, "%s ilv%d no dependency", v.name, ilv);
            timeit([&] { hipLaunchKernelGGL(v.fn, dim3(NCU), dim3(256), lds, st, dop, n_ops, x0, xg, sy, 1 | b); }, nm, tmp);
            snprintf(nm, sizeof nm, "%s ilv%d ENGINE (dependent chain)", v.name, ilv);
            timeit([&] { hipLaunchKernelGGL(v.fn, dim3(NCU), dim3(256), lds, st, dop, n_ops, x0, xg, sy, b); }, nm, got);
            if (!ilv) {
                size_t bad = 0; for (int i = 0; i < RS[(n_ops - 1) % 4]; ++i) if (ref[i] != got[i]) ++bad;
                printf("    engine vs launches: %zu of %d outputs differ\n", bad, RS[(n_ops - 1) % 4]);
            }
        }
    }
    return 0;
}
