// glds_rate_probe.hip — how fast can one CU pull L2-resident data through (a) global_load_lds_dwordx4
// (LDS-DMA) and (b) global_load_dwordx4 into VGPRs, for a coalesced (1 KiB per instruction) and a
// "fragment-shaped" (32 rows x 32 B per instruction) address pattern?  Decides how the prefill GEMM
// stages its operands (thk_prefill.hip).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 glds_rate_probe.hip -o glds_rate_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int AUX = 0>
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, AUX);
}
// region: 2 MiB shared by all blocks (L2 resident).  FRAG: lane -> row (lane&31) * 8192 B + (lane>>5)*16 + k*32
template <bool DMA, int FRAG>
__global__ __launch_bounds__(256, 1) void probe(const char* __restrict__ src, int iters, float* out) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f4 acc = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        // 16 instructions per wave per iteration = 16 KiB per wave, 64 KiB per block
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int slot = (it * 16 + i) & 31;               // walk 32 KiB windows so that consecutive iterations differ
            const char* p;
            if (FRAG == 1) p = src + (size_t)((wave * 32 + (lane & 31)) * 8192) + ((slot * 32 + (lane >> 5) * 16) & 8191) + (size_t)(blockIdx.x & 1) * (1 << 20);
            else if (FRAG == 2) p = src + (size_t)((wave * 16 + (lane >> 2)) * 8192) + ((slot * 64 + (lane & 3) * 16) & 8191) + (size_t)(blockIdx.x & 1) * (1 << 20);          // 16 rows x 64 B, quad-contiguous
            else if (FRAG == 3) p = src + (size_t)((wave * 16 + (lane >> 2)) * 8192) + ((slot * 64 + ((lane & 3) ^ ((lane >> 4) & 3)) * 16) & 8191) + (size_t)(blockIdx.x & 1) * (1 << 20);   // same, pieces permuted inside the quad
            else if (FRAG == 4) p = src + (size_t)((wave * 32 + (lane >> 1)) * 8192) + ((slot * 32 + (lane & 1) * 16) & 8191) + (size_t)(blockIdx.x & 1) * (1 << 20);          // 32 rows x 32 B, pair-contiguous
            else if (FRAG == 5) p = src + (size_t)((wave * 8 + (lane >> 3)) * 8192) + ((slot * 128 + (lane & 7) * 16) & 8191) + (size_t)(blockIdx.x & 1) * (1 << 20);         // 8 rows x 128 B
            else p = src + (size_t)(((blockIdx.x & 7) * 4 + wave) * 32 + slot) * 1024 + lane * 16;
            if (DMA) glds16(p, lds + (wave * 16 + i) * 1024);
            else acc += *reinterpret_cast<const f4*>(p);
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
    if (DMA) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); acc = *reinterpret_cast<const f4*>(lds + threadIdx.x * 16); }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

// HBM streaming in the prefill GEMM's shape: a block owns 256 rows of 8 KiB (4 waves x 64 rows) and walks along the
// rows; per step every wave issues 4 instructions that fetch PIECE bytes from each of (4096 / PIECE) rows... i.e.
// PIECE = 64: 16 rows x 64 B per instruction (KC = 32 halfs), 128: 8 rows x 128 B, 256: 4 rows x 256 B, 1024: 1 row.
template <int PIECE, int AUX = 0>
__global__ __launch_bounds__(256, 1) void hbm_probe(const char* __restrict__ src, int row_blocks, float* out) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int LPR = PIECE / 16, RPI = 64 / LPR;          // lanes per row, rows per instruction
    constexpr int IPS = 64 / RPI;                            // instructions per wave to cover its 64 rows once
    int n = 0;
    for (int rb = 0; rb < row_blocks; ++rb) {
        const char* base = src + ((size_t)(rb * gridDim.x + blockIdx.x) * 256 + wave * 64) * 8192;
        for (int col = 0; col < 8192; col += PIECE) {
#pragma unroll
            for (int j = 0; j < IPS; ++j) {
                const char* p = base + (size_t)(j * RPI + lane / LPR) * 8192 + col + (lane % LPR) * 16;
                glds16<AUX>(p, lds + ((n & 15) * 4 + wave) * 1024);
                ++n;
            }
            if (IPS >= 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
    f4 acc = *reinterpret_cast<const f4*>(lds + threadIdx.x * 16);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
template <int PIECE, int AUX = 0> static void run_hbm(const char* src, float* out) {
    const int rbs = 4;     // 256 blocks x 4 x 2 MiB = 2 GiB
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute((const void*)hbm_probe<PIECE, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipLaunchKernelGGL((hbm_probe<PIECE, AUX>), dim3(256), dim3(256), 65536, 0, src, 1, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((hbm_probe<PIECE, AUX>), dim3(256), dim3(256), 65536, 0, src, rbs, out);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 256.0 * rbs * 256 * 8192;
    printf("HBM stream (aux=%d), %4d B per row per step: %8.1f GB/s total (%.2f ms for %.0f MiB)\n", AUX, PIECE, bytes / ms / 1e6, ms, bytes / 1048576);
}

template <bool DMA, int FRAG> static void run(const char* name, const char* src, float* out, int blocks) {
    const int iters = 2000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute((const void*)probe<DMA, FRAG>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipLaunchKernelGGL((probe<DMA, FRAG>), dim3(blocks), dim3(256), 65536, 0, src, 50, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<DMA, FRAG>), dim3(blocks), dim3(256), 65536, 0, src, iters, out);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)blocks * iters * 65536;
    printf("%-34s blocks=%3d  %8.1f GB/s total  %7.1f GB/s per CU  (%.1f B/clk/CU @2.4GHz)\n", name, blocks, bytes / ms / 1e6, bytes / ms / 1e6 / blocks, bytes / ms / 1e6 / blocks / 2.4);
}
int main() {
    char* src; float* out;
    CHECK(hipMalloc(&src, 4 << 20)); CHECK(hipMemset(src, 1, 4 << 20)); CHECK(hipMalloc(&out, 64));
    for (int blocks : {256}) {
        run<true, 0>("glds x4, coalesced 1 KiB/instr", src, out, blocks);
        run<false, 0>("global_load x4 -> VGPR, coalesced", src, out, blocks);
        run<true, 1>("glds x4, fragment (32 rows x 2x16 B)", src, out, blocks);
        run<false, 1>("global_load x4 -> VGPR, fragment", src, out, blocks);
        run<true, 2>("glds x4, 16 rows x 64 B (quads)", src, out, blocks);
        run<true, 3>("glds x4, 16 rows x 64 B permuted", src, out, blocks);
        run<true, 4>("glds x4, 32 rows x 32 B (pairs)", src, out, blocks);
        run<true, 5>("glds x4, 8 rows x 128 B", src, out, blocks);
        run<false, 2>("global_load x4, 16 rows x 64 B", src, out, blocks);
    }
    char* big; CHECK(hipMalloc(&big, (size_t)2 << 30)); CHECK(hipMemset(big, 1, (size_t)2 << 30));
    run_hbm<64>(big, out); run_hbm<128>(big, out); run_hbm<256>(big, out); run_hbm<1024>(big, out);
    run_hbm<64, 2>(big, out); run_hbm<128, 2>(big, out); run_hbm<256, 2>(big, out); run_hbm<1024, 2>(big, out);
    run_hbm<64>(big, out); run_hbm<128>(big, out); run_hbm<256>(big, out); run_hbm<1024>(big, out);
    run_hbm<64, 2>(big, out); run_hbm<128, 2>(big, out); run_hbm<256, 2>(big, out); run_hbm<1024, 2>(big, out);
    return 0;
}
