// read_floor_probe.hip — what does a LLaMA-7B decode step cost on this stack when its 161 launches do NOTHING but read their
// operand?  Every launch of the decode graph is replaced by a bare streaming read of the same number of bytes from
// the same kind of place (a 13.75 GB slab walked front to back once per step, so nothing survives in L2 / the memory-side cache
// from one step to the next), `global_load_dwordx4 nt`, a constant number of KiB in flight per wave and no arithmetic beyond an
// XOR that keeps the loads alive.  The result is the floor of ANY launch-per-operation decode on this runtime: bytes / HBM rate +
// what 161 dependent kernel boundaries cost - and, swept over geometry, which geometry a streaming kernel should have.
//   sections (second argument, default "cfks"):
//   c  chain     the 161 launches in stream order (what the decode graph is), product geometry; the same nodes without
//                dependency edges; one launch that reads the whole slab
//   f  geometry  the chain with every launch at b workgroups/CU x t threads, k KiB in flight per wave, chunk size G
//   k  per kind  160 launches of one kind back to back on rotating operands, swept over (workgroups/CU, KiB in flight, chunk)
//   s  sizes     time vs bytes: intercept = per-launch cost, slope = stream rate
// A wave's stream: the launch's 1 KiB slots (64 lanes x 16 B) are cut into chunks of G slots; wave w takes chunks w, w + W, ...
// (G = 1: all waves sweep the operand together; G = 16: a row pair of 4096 f16 columns, the product's unit; G = 0: one contiguous
// span per wave) and keeps U slots in flight from its first load to its last (consume slot i, request slot i + U).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 read_floor_probe.hip -o read_floor_probe     Run: ./read_floor_probe [replays] [sections]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int U, int MAXT>
__global__ __launch_bounds__(MAXT) void read_kernel(const u4* __restrict__ p, int G, long total_slots, unsigned* __restrict__ sink, int rot_mul) {
    const int lane = threadIdx.x & 63;
    const long W = (long)gridDim.x * (blockDim.x >> 6);
    const long wave = __builtin_amdgcn_readfirstlane((int)((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (G <= 0) G = (int)((total_slots + W - 1) / W);            // one contiguous span per wave
    const long nchunks = (total_slots + G - 1) / G;
    if (wave >= nchunks) return;
    const long mine = (nchunks - wave + W - 1) / W;              // chunks of this wave
    long n = mine * G;                                           // its slots (the operand's last chunk may be short)
    if (wave + (mine - 1) * W == nchunks - 1) n -= nchunks * G - total_slots;
    // issue cursor (all scalar, branch-free: a conditional load makes hipcc wait for vmcnt(0) at the join).  Past the wave's last
    // slot the cursor stays on it: a launch whose per-wave slot count is no multiple of U re-requests that slot up to U - 1 times
    // (the same 1 KiB the wave has just fetched), the requests the memory system sees beyond the operand are those.
    // rot_mul != 0: the wave starts every chunk at slot (wave * rot_mul) % G and wraps around inside the chunk (same bytes, same
    // owner - only the ORDER differs from wave to wave, so the waves of a launch do not walk their chunks in step)
    const int rot = rot_mul ? (int)((wave * rot_mul) % G) : 0;
    long ichunk = wave; int ij = 0; long issued = 0;
    auto next_addr = [&]() -> const u4* {
        int je = ij + rot; je = je >= G ? je - G : je;
        const u4* a = p + (ichunk * G + je) * 64 + lane;
        const bool more = issued + 1 < n;
        const bool wrap = ij + 1 == G;
        const long nchunk = wrap ? ichunk + W : ichunk;
        const int nij = wrap ? 0 : ij + 1;
        ichunk = more ? nchunk : ichunk; ij = more ? nij : ij; issued += more ? 1 : 0;
        return a;
    };
    u4 r[U];
    unsigned acc = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = __builtin_nontemporal_load(next_addr());
    const long rounds = n > U ? (n - U + U - 1) / U : 0;
    for (long b = 0; b < rounds; ++b) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc ^= r[u].x ^ r[u].y ^ r[u].z ^ r[u].w;
            r[u] = __builtin_nontemporal_load(next_addr());
            __builtin_amdgcn_sched_barrier(0);                  // request i + U stays right behind consume i
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= r[u].x ^ r[u].y ^ r[u].z ^ r[u].w;
    asm volatile("" ::"v"(acc));                              // the loads stay alive, nothing is written
    if (total_slots < 0) sink[0] = acc;
}
__global__ void fill_kernel(unsigned* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = x;
    }
}

typedef void (*read_fn)(const u4*, int, long, unsigned*, int);
static read_fn kernel_for(int U, int threads) {
    if (threads > 256) return U == 2 ? read_kernel<2, 1024> : U == 4 ? read_kernel<4, 1024> : U == 8 ? read_kernel<8, 1024> : read_kernel<16, 1024>;
    return U == 2 ? read_kernel<2, 256> : U == 4 ? read_kernel<4, 256> : U == 8 ? read_kernel<8, 256> : U == 16 ? read_kernel<16, 256> : read_kernel<32, 256>;
}
struct Launch { const char* kind; size_t bytes; int blocks; int threads; size_t off; int G; };
struct Geo { int U = 16; int blocks = 0; int threads = 0; int G = -1; int rot = 0; };     // 0 / -1: what the launch itself says
static void launch(const Launch& L, const char* pool, const Geo& geo, unsigned* sink, hipStream_t st) {
    const int blocks = geo.blocks ? geo.blocks : L.blocks, threads = geo.threads ? geo.threads : L.threads;
    const long slots = (long)((L.bytes + 1023) / 1024);
    const u4* p = reinterpret_cast<const u4*>(pool + L.off);
    hipLaunchKernelGGL(kernel_for(geo.U, threads), dim3(blocks), dim3(threads), 0, st, p, geo.G >= 0 ? geo.G : L.G, slots, sink, geo.rot);
}
static double replay_ms(hipGraphExec_t x, hipStream_t st, int replays) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) CHECK(hipGraphLaunch(x, st));
    CHECK(hipStreamSynchronize(st));
    std::vector<float> t;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a, st));
        for (int i = 0; i < replays; ++i) CHECK(hipGraphLaunch(x, st));
        CHECK(hipEventRecord(b, st));
        CHECK(hipStreamSynchronize(st));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms / replays);
    }
    std::sort(t.begin(), t.end());
    CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    return t[t.size() / 2];
}
static hipGraphExec_t capture_chain(const std::vector<Launch>& ls, const char* pool, const Geo& geo, unsigned* sink, hipStream_t st, int steps = 1) {
    hipGraph_t g; hipGraphExec_t x;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int s = 0; s < steps; ++s) for (const Launch& L : ls) launch(L, pool, geo, sink, st);
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
    CHECK(hipGraphDestroy(g));
    return x;
}
static double chain_ms(const std::vector<Launch>& ls, const char* pool, const Geo& geo, unsigned* sink, hipStream_t st, int replays, int steps = 1) {
    hipGraphExec_t x = capture_chain(ls, pool, geo, sink, st, steps);
    const double ms = replay_ms(x, st, replays) / steps;
    CHECK(hipGraphExecDestroy(x));
    return ms;
}
// the same kernel nodes with no edges between them
static hipGraphExec_t build_free(const std::vector<Launch>& ls, const char* pool, unsigned* sink) {
    hipGraph_t g; hipGraphExec_t x;
    CHECK(hipGraphCreate(&g, 0));
    std::vector<const u4*> ps(ls.size()); std::vector<int> gs(ls.size()); std::vector<long> slots(ls.size());
    std::vector<void*> argv(ls.size() * 5); static int zero = 0;
    for (size_t i = 0; i < ls.size(); ++i) {
        const Launch& L = ls[i];
        slots[i] = (long)((L.bytes + 1023) / 1024); gs[i] = L.G;
        ps[i] = reinterpret_cast<const u4*>(pool + L.off);
        argv[5 * i + 0] = &ps[i]; argv[5 * i + 1] = &gs[i]; argv[5 * i + 2] = &slots[i]; argv[5 * i + 3] = &sink; argv[5 * i + 4] = &zero;
        hipKernelNodeParams kp{};
        kp.func = reinterpret_cast<void*>(kernel_for(16, L.threads));
        kp.gridDim = dim3(L.blocks); kp.blockDim = dim3(L.threads); kp.sharedMemBytes = 0; kp.kernelParams = &argv[5 * i]; kp.extra = nullptr;
        hipGraphNode_t node;
        CHECK(hipGraphAddKernelNode(&node, g, nullptr, 0, &kp));
    }
    CHECK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
    CHECK(hipGraphDestroy(g));
    return x;
}

int main(int argc, char** argv) {
    const int replays = argc > 1 ? atoi(argv[1]) : 20;
    const char* sections = argc > 2 ? argv[2] : "cfks";
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int CU = prop.multiProcessorCount;
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    // LLaMA-7B, T = 512: algorithmic bytes per launch, the product's workgroup counts and row-group sizes (auto_geometry())
    const size_t E = 4096, F = 11008, V = 32000, T = 512;
    const int LAYERS = 32;
    struct Kind { const char* name; size_t bytes; int blocks, threads, G; double product_us; };
    const Kind kinds[6] = {
        // workgroups / threads / chunk (KiB a wave reads before it jumps) / rocprofv3 us of the product WHEN THE PROBE WAS WRITTEN (row pairs for
        // qkv and w13, whole w2 rows, 3 / 4 / 2 workgroups per CU: profiles/r04_kernel_stats_final_binary.csv).  The product the probe
        // led to reads 8 KiB chunks at one workgroup per CU: 16.11 / 5.48 / 7.30 / 27.09 / 14.90 / 41.37 us (r04_kernel_stats_final.csv)
        {"norm_qkv_rope_kv", 3 * E * E * 2, 768, 256, 16, 17.11}, {"attn_decode", 2 * T * E * 4, 128, 512, 16, 5.49}, {"attn_wo_resid", E * E * 2, 256, 256, 8, 8.41},
        {"norm_w13_swiglu", 2 * E * F * 2, 1024, 256, 16, 27.55}, {"w2_resid", E * F * 2, 512, 256, 43, 15.98}, {"norm_lmhead", V * E * 2, 2048, 256, 8, 41.60}};
    std::vector<Launch> step;
    size_t off = 0, step_bytes = 0;
    auto push = [&](const Kind& k) { step.push_back({k.name, k.bytes, k.blocks, k.threads, off, k.G}); off += (k.bytes + 4095) / 4096 * 4096; step_bytes += k.bytes; };
    for (int l = 0; l < LAYERS; ++l) for (int k = 0; k < 5; ++k) push(kinds[k]);
    push(kinds[5]);
    const size_t pool_bytes = off;
    char* pool; unsigned* sink;
    CHECK(hipMalloc((void**)&pool, pool_bytes)); CHECK(hipMalloc((void**)&sink, 64));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, reinterpret_cast<unsigned*>(pool), pool_bytes / 4);
    CHECK(hipStreamSynchronize(st));
    printf("# read_floor_probe on %s (%d CUs): LLaMA-7B decode step at T = 512 as %zu bare streaming reads, %.3f GB per step, %d replays x 5, median\n",
           prop.gcnArchName, CU, step.size(), step_bytes / 1e9, replays);
    printf("# product on the same pool of boxes (bench.py): 2.43-2.46 ms per step when the probe was written, 2.31-2.34 ms after its tables were built into the kernels (DESIGN.md 4.8)\n");
    auto report = [&](const char* what, double ms) { printf("%-76s %8.4f ms/step  %6.3f TB/s  %5.1f %%\n", what, ms, step_bytes / ms / 1e9, step_bytes / ms / 1e9 / 8.0 * 100); fflush(stdout); };
    char buf[160];

    if (strchr(sections, 'c')) {
        for (int U : {2, 4, 8, 16, 32}) {
            Geo geo; geo.U = U;
            snprintf(buf, sizeof buf, "chain, product geometry and row groups, %2d KiB in flight per wave", U);
            report(buf, chain_ms(step, pool, geo, sink, st, replays));
        }
        for (int G : {4, 8, 16, 32}) for (int U : {8, 16}) {      // what the chunk size alone is worth: the product's workgroup counts, every launch cut into chunks of G KiB
            Geo geo; geo.U = U; geo.G = G;
            snprintf(buf, sizeof buf, "chain, product workgroup counts, chunks of %2d KiB, %2d KiB in flight per wave", G, U);
            report(buf, chain_ms(step, pool, geo, sink, st, replays));
        }
        { Geo geo; report("chain, product geometry, 16 KiB in flight, 8 steps per graph", chain_ms(step, pool, geo, sink, st, std::max(1, replays / 4), 8)); }
        {
            hipGraphExec_t x = build_free(step, pool, sink);
            report("free: the same 161 kernel nodes without dependency edges", replay_ms(x, st, replays));
            CHECK(hipGraphExecDestroy(x));
        }
        for (int U : {4, 16}) {   // one launch of everything: no boundary at all
            std::vector<Launch> one{{"all", step_bytes, 8 * CU, 256, 0, 16}};
            Geo geo; geo.U = U;
            snprintf(buf, sizeof buf, "one launch that reads the whole slab (8 workgroups/CU, chunks of 16, %2d KiB in flight)", U);
            report(buf, chain_ms(one, pool, geo, sink, st, replays));
        }
    }
    if (strchr(sections, 'f')) {
        printf("# geometry: every launch of the chain at b workgroups/CU x t threads; rows = chunk size G (0 = one contiguous span per wave), columns = KiB in flight per wave 2 4 8 16; ms per step\n");
        for (int threads : {256, 512, 1024}) for (int bpc : {1, 2, 3, 4, 5, 6, 8}) {
            if (bpc * threads > 2048) continue;
            for (int G : {1, 8, 16, 0}) {
                printf("b=%d t=%4d G=%2d :", bpc, threads, G);
                for (int U : {2, 4, 8, 16}) {
                    if (threads > 256 && U > 8 && bpc * threads > 1024) { printf("       -"); continue; }
                    Geo geo; geo.U = U; geo.blocks = bpc * CU; geo.threads = threads; geo.G = G;
                    printf(" %7.4f", chain_ms(step, pool, geo, sink, st, std::max(4, replays / 2)));
                }
                printf("\n"); fflush(stdout);
            }
        }
    }
    if (strchr(sections, 'k')) {
        printf("# per kind: 160 launches back to back on rotating operands; us per launch; rows = (waves/CU, threads per workgroup, chunk G; G = 0: one contiguous span per wave), columns = KiB in flight per wave 2 4 8 16\n");
        for (int k = 0; k < 6; ++k) {
            std::vector<Launch> ls;
            const size_t stride = (kinds[k].bytes + 4095) / 4096 * 4096;
            const size_t slots_avail = pool_bytes / stride;
            for (int i = 0; i < 160; ++i) ls.push_back({kinds[k].name, kinds[k].bytes, kinds[k].blocks, kinds[k].threads, (i % slots_avail) * stride, kinds[k].G});
            Geo geo;
            const double us0 = chain_ms(ls, pool, geo, sink, st, std::max(2, replays / 4)) * 1e3 / 160;
            printf("%s: %.2f MB; product %.2f us (rocprofv3); bare read in the product's geometry (%d x %d, G = %d, 16 KiB): %.2f us = %.3f TB/s\n", kinds[k].name, kinds[k].bytes / 1e6,
                   kinds[k].product_us, kinds[k].blocks, kinds[k].threads, kinds[k].G, us0, kinds[k].bytes / us0 / 1e6);
            double best = 1e9; char bestcfg[64] = "";
            for (int threads : {256, 512}) for (int wpc : {4, 8, 12, 16, 24, 32}) {       // waves per CU
                if ((wpc * 64) % threads) continue;
                const int bpc = wpc * 64 / threads;
                for (int G : {1, 2, 4, 8, 16, kinds[k].G > 16 ? kinds[k].G : 0}) {
                    printf("  waves/CU=%2d t=%3d G=%2d :", wpc, threads, G);
                    for (int U : {2, 4, 8, 16}) {
                        Geo g2; g2.U = U; g2.blocks = bpc * CU; g2.threads = threads; g2.G = G;
                        const double us = chain_ms(ls, pool, g2, sink, st, std::max(2, replays / 4)) * 1e3 / 160;
                        printf(" %7.2f", us);
                        if (us < best) { best = us; snprintf(bestcfg, sizeof bestcfg, "waves/CU=%d t=%d G=%d %d KiB", wpc, threads, G, U); }
                    }
                    printf("\n"); fflush(stdout);
                }
            }
            printf("  best %.2f us = %.3f TB/s at %s; product / best bare = %.3f\n", best, kinds[k].bytes / best / 1e6, bestcfg, kinds[k].product_us / best);
        }
    }
    if (strchr(sections, 'r')) {
        printf("# rotation: the product's geometry (workgroups, threads) per kind; rows = chunk G x start-slot rule (0: every wave starts its chunk at slot 0; 1: at slot wave %% G; 4: at slot 4 * wave %% G); columns = KiB in flight 8 16; us per launch\n");
        for (int k = 0; k < 6; ++k) {
            std::vector<Launch> ls;
            const size_t stride = (kinds[k].bytes + 4095) / 4096 * 4096;
            const size_t slots_avail = pool_bytes / stride;
            for (int i = 0; i < 160; ++i) ls.push_back({kinds[k].name, kinds[k].bytes, kinds[k].blocks, kinds[k].threads, (i % slots_avail) * stride, kinds[k].G});
            printf("%s (product %.2f us)\n", kinds[k].name, kinds[k].product_us);
            const long slots = (long)(kinds[k].bytes / 1024);
            for (int G : {4, 8, 16, 32, kinds[k].G > 16 ? kinds[k].G : 64}) {
                if (slots % G) continue;
                for (int rot : {0, 1, 4}) {
                    printf("  G=%2d rot=%d :", G, rot);
                    for (int U : {8, 16}) {
                        Geo g2; g2.U = U; g2.G = G; g2.rot = rot;
                        printf(" %7.2f", chain_ms(ls, pool, g2, sink, st, std::max(2, replays / 4)) * 1e3 / 160);
                    }
                    printf("\n"); fflush(stdout);
                }
            }
        }
        printf("# the whole chain in the product's geometry, 16 KiB in flight: product chunks, start-slot rule 0 | 1 | 4\n");
        for (int rot : {0, 1, 4}) {
            Geo geo; geo.rot = rot;
            snprintf(buf, sizeof buf, "chain, product geometry and row groups, start-slot rule %d", rot);
            report(buf, chain_ms(step, pool, geo, sink, st, replays));
        }
    }
    if (strchr(sections, 's')) {
        for (int U : {4, 16}) {
            printf("# time vs bytes (4 workgroups/CU x 256 threads, chunks of 16, %d KiB in flight, 160 launches on rotating operands)\n", U);
            double xs[6], ys[6]; int np = 0;
            for (size_t mb : {8, 16, 32, 64, 128, 256}) {
                std::vector<Launch> ls;
                const size_t bytes = mb << 20;
                const size_t slots_avail = pool_bytes / bytes;
                for (int i = 0; i < 160; ++i) ls.push_back({"sweep", bytes, 4 * CU, 256, (i % slots_avail) * bytes, 16});
                Geo geo; geo.U = U;
                const double us = chain_ms(ls, pool, geo, sink, st, std::max(2, replays / 4)) * 1e3 / 160;
                printf("  %4zu MiB  %8.2f us  %6.3f TB/s\n", mb, us, bytes / us / 1e6);
                xs[np] = bytes / 1e6; ys[np] = us; ++np;
            }
            double sx = 0, sy = 0, sxx = 0, sxy = 0;
            for (int i = 0; i < np; ++i) { sx += xs[i]; sy += ys[i]; sxx += xs[i] * xs[i]; sxy += xs[i] * ys[i]; }
            const double slope = (np * sxy - sx * sy) / (np * sxx - sx * sx), icpt = (sy - slope * sx) / np;
            printf("# least squares: %.2f us per launch + bytes / %.3f TB/s\n", icpt, 1.0 / slope);
        }
    }
    return 0;
}
