// clock_probe: what does s_memtime tick at, and what shader clock does an MFMA-bound grid sustain on MI355X?
// Each wave times N back-to-back v_mfma_f32_32x32x16_f16 (4 independent accumulators) with s_memtime (shader cycles) and
// wall_clock64 (100 MHz).  Build: hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 1) void k(int n, unsigned long long* out, float* sink) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = w1 - w0; }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) sink[0] = c0[0];
}
int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 4 * 2 * 8); hipMalloc(&sink, 64);
    for (int blocks : {1, 256}) for (int n : {2000, 20000}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, n, out, sink); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, n, out, sink); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 8); hipMemcpy(h.data(), out, blocks * 8 * 8, hipMemcpyDeviceToHost);
        double st = 0, sw = 0; for (int i = 0; i < blocks * 4; ++i) { st += h[2 * i]; sw += h[2 * i + 1]; }
        st /= blocks * 4; sw /= blocks * 4;
        printf("blocks %3d  n %6d : %.0f memtime ticks, %.0f wall ticks (x10 ns) per wave -> %.3f ticks/ns; %.2f memtime ticks per MFMA; event time %.1f us -> %.2f ns per MFMA\n",
               blocks, n, st, sw, st / (sw * 10.0), st / (4.0 * n), ms * 1e3, ms * 1e6 / (4.0 * n));
    }
    return 0;
}
