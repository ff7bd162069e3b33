// thk_prefill.hip — batched prompt-prefill GEMM on the CDNA4 matrix cores (config C3).
//
//   Y[M,R] = X[M,C] * W[R,C]^T     X,Y f32 row-major, W GGML f16 row-major.
//
// This is the one place on the path where the work is a real dense contraction
// (M = 128 prompt tokens share every weight row), so it goes to MFMA
// (v_mfma_f32_32x32x16_f16) instead of the HBM-streaming mat-vec.  The reference
// multiplies f32 activations by f16-decoded weights in f32 (cmdbuf_mat_mul with f16 B,
// th.cpp:396-539, batch branch th-llama.cpp:307-311); to keep that precision on f16
// matrix cores the activations are split x = hi + lo (both f16, |lo| <= 2^-11 |x|) and
// two MFMAs accumulate into the same f32 tile: error ~2^-22 relative, f32-class.
//
// Fragment layout used (gfx950 32x32x16, 8 f16 per lane per operand):
//   A[i = lane&31][k = 8*(lane>>5) + e], B[k = 8*(lane>>5) + e][j = lane&31], e = 0..7
//   D[row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = lane&31], reg = 0..15
// We set A = 32 weight rows, B = 32 tokens, so both operands are 16-byte contiguous
// loads along K straight from row-major memory (no transposes), and each lane ends up
// with 4 consecutive output features of one token => float4 stores.
//
// v1 structure (round 1): one wave per workgroup, tile = 32 weight rows x (32*MT) tokens,
// W fragments streamed from HBM (each element read exactly once), X fragments served by
// L1/L2 (X hi/lo is <= 5.6 MB and shared by all workgroups).  LDS staging of X and a
// split-K / multi-wave schedule are the next steps.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "thk_kernels.hpp"

namespace thk {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// X f32 [M,C] -> hi,lo f16 [Mpad,C] (rows >= M zero-filled so MFMA tiles need no masking)
__global__ void split_hi_lo_kernel(const float* __restrict__ X, int M, int Mpad, int C, _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
    const size_t n = (size_t)Mpad * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / C;
        float x = row < (size_t)M ? X[i] : 0.f;
        const _Float16 h = (_Float16)x;
        hi[i] = h;
        lo[i] = (_Float16)(x - (float)h);
    }
}

// One wave: rows [r0, r0+32) of W, tokens [0, 32*MT).
template <int MT>
__global__ __launch_bounds__(64) void gemm_prefill_kernel(const uint16_t* __restrict__ Wp, int R, int C,
                                                          const _Float16* __restrict__ Xhi, const _Float16* __restrict__ Xlo,
                                                          int M, float* __restrict__ Y) {
    const int lane = threadIdx.x;
    const int r0 = blockIdx.x * 32;
    const int li = lane & 31, kh = (lane >> 5) * 8;
    int wrow = r0 + li; if (wrow >= R) wrow = R - 1;          // clamp (results of clamped rows are not stored)
    const _Float16* wp = reinterpret_cast<const _Float16*>(Wp) + (size_t)wrow * C + kh;
    const _Float16* xh = Xhi + (size_t)li * C + kh;
    const _Float16* xl = Xlo + (size_t)li * C + kh;

    f16v acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    for (int k0 = 0; k0 < C; k0 += 32) {       // two K=16 steps per iteration
        const h8 a0 = __builtin_nontemporal_load(reinterpret_cast<const h8*>(wp + k0));
        const h8 a1 = __builtin_nontemporal_load(reinterpret_cast<const h8*>(wp + k0 + 16));
        h8 bh0[MT], bl0[MT], bh1[MT], bl1[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const size_t off = (size_t)t * 32 * C + k0;
            bh0[t] = *reinterpret_cast<const h8*>(xh + off); bl0[t] = *reinterpret_cast<const h8*>(xl + off);
            bh1[t] = *reinterpret_cast<const h8*>(xh + off + 16); bl1[t] = *reinterpret_cast<const h8*>(xl + off + 16);
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bh0[t], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bl0[t], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bh1[t], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bl1[t], acc[t], 0, 0, 0);
        }
    }
    // D[row -> weight row][col -> token]; lane holds token (lane&31), 4 groups of 4 consecutive rows
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int tok = t * 32 + li;
        if (tok < M) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = r0 + 8 * g + 4 * (lane >> 5);
                float* dst = Y + (size_t)tok * R + r;
                if (r + 3 < R) {
                    *reinterpret_cast<f4*>(dst) = f4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
                } else {
                    for (int e = 0; e < 4; ++e) if (r + e < R) dst[e] = acc[t][4 * g + e];
                }
            }
        }
    }
}

// ---------------------------------------------------------------- v2: LDS-staged X, 4 waves, split-K
// Workgroup = 4 waves = 128 weight rows x (32*MT) tokens; the four waves share one X tile (hi and
// lo, 64 columns at a time, double-buffered in LDS with a 16-byte row pad => conflict-free
// ds_read_b128), each wave streams its own 32 weight rows straight from HBM.  K is split across
// gridDim.y workgroups so that small matrices still fill the chip; partial tiles go to a workspace
// and are summed in a fixed order by reduce_splits_kernel (deterministic, no atomics).
constexpr int kKC = 32;                  // K columns per LDS stage (2 MFMA k-steps)
constexpr int kXS = kKC + 8;             // LDS row stride in halfs (80 B = 5 x 16-byte slots: conflict-free)
constexpr int kKSteps = kKC / 16;

template <int MT>
__global__ __launch_bounds__(256) void gemm_prefill_v2_kernel(const uint16_t* __restrict__ Wp, int R, int C,
                                                              const _Float16* __restrict__ Xhi, const _Float16* __restrict__ Xlo,
                                                              int M, float* __restrict__ Yp, int chunks_per_split) {
    __shared__ __attribute__((aligned(16))) _Float16 sh[2][2][MT * 32][kXS];   // [stage][hi/lo][token][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = (lane >> 5) * 8;
    const int r0 = blockIdx.x * 128 + wave * 32;
    int wrow = r0 + li; if (wrow >= R) wrow = R - 1;
    const _Float16* wp = reinterpret_cast<const _Float16*>(Wp) + (size_t)wrow * C + kh;
    const int nchunks = C / kKC;
    const int c_begin = blockIdx.y * chunks_per_split, c_end = min(nchunks, c_begin + chunks_per_split);

    f16v acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // X tile loader: (MT*32 rows) x (kKC/8) 16-byte segments per array, spread over the 256 threads
    constexpr int SPR = kKC / 8;                       // segments per row
    constexpr int NSEG = MT * 32 * SPR;                // segments per array
    constexpr int SEG = (NSEG + 255) / 256;            // per thread
    h8 xr[2][SEG];
    auto load_x = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < SEG; ++i) {
            const int idx = min(tid + i * 256, NSEG - 1), row = idx / SPR, seg = idx % SPR;
            const size_t off = (size_t)row * C + (size_t)chunk * kKC + seg * 8;
            xr[0][i] = *reinterpret_cast<const h8*>(Xhi + off);
            xr[1][i] = *reinterpret_cast<const h8*>(Xlo + off);
        }
    };
    auto store_x = [&](int stage) {
#pragma unroll
        for (int i = 0; i < SEG; ++i) {
            const int idx = tid + i * 256, row = idx / SPR, seg = idx % SPR;
            if (idx < NSEG) {
                *reinterpret_cast<h8*>(&sh[stage][0][row][seg * 8]) = xr[0][i];
                *reinterpret_cast<h8*>(&sh[stage][1][row][seg * 8]) = xr[1][i];
            }
        }
    };
    h8 wr[kKSteps], wn[kKSteps];
    auto load_w = [&](int chunk, h8 (&w)[kKSteps]) {
#pragma unroll
        for (int ks = 0; ks < kKSteps; ++ks) w[ks] = __builtin_nontemporal_load(reinterpret_cast<const h8*>(wp + (size_t)chunk * kKC + ks * 16));
    };

    if (c_begin < c_end) {
        load_x(c_begin); load_w(c_begin, wr);
        store_x(0);
        __syncthreads();
        for (int c = c_begin; c < c_end; ++c) {
            const int stage = (c - c_begin) & 1;
            const bool more = c + 1 < c_end;
            if (more) { load_x(c + 1); load_w(c + 1, wn); }      // next stage in flight during the MFMAs
#pragma unroll
            for (int ks = 0; ks < kKSteps; ++ks) {
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const h8 bh = *reinterpret_cast<const h8*>(&sh[stage][0][t * 32 + li][ks * 16 + kh]);
                    const h8 bl = *reinterpret_cast<const h8*>(&sh[stage][1][t * 32 + li][ks * 16 + kh]);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[ks], bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[ks], bl, acc[t], 0, 0, 0);
                }
            }
            if (more) {
                store_x(stage ^ 1);                              // the other buffer was last read one iteration ago
#pragma unroll
                for (int ks = 0; ks < kKSteps; ++ks) wr[ks] = wn[ks];
            }
            __syncthreads();
        }
    }
    float* Y = Yp + (size_t)blockIdx.y * M * R;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int tok = t * 32 + li;
        if (tok < M) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = r0 + 8 * g + 4 * (lane >> 5);
                float* dst = Y + (size_t)tok * R + r;
                if (r + 3 < R) {
                    *reinterpret_cast<f4*>(dst) = f4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
                } else {
                    for (int e = 0; e < 4; ++e) if (r + e < R) dst[e] = acc[t][4 * g + e];
                }
            }
        }
    }
}

__global__ void reduce_splits_kernel(const float* __restrict__ part, int ks, size_t n4, float* __restrict__ Y) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f4 s = reinterpret_cast<const f4*>(part)[i];
    for (int k = 1; k < ks; ++k) s += reinterpret_cast<const f4*>(part)[(size_t)k * n4 + i];
    reinterpret_cast<f4*>(Y)[i] = s;
}

static int prefill_splits(int R, int C) {
    const int rb = (R + 127) / 128, nchunks = C / kKC;
    int ks = 1;
    while (ks < 8 && rb * ks < 256 && nchunks / (ks * 2) >= 8) ks *= 2;
    return ks;
}

size_t gemm_prefill_workspace_bytes(int M, int R, int C) {
    const int mc = M < 128 ? M : 128;
    const int Mpad = (mc + 31) / 32 * 32;
    size_t b = ((size_t)Mpad * C * 2 * 2 + 255) / 256 * 256;
    if (C % kKC == 0) b += (size_t)prefill_splits(R, C) * mc * R * 4;
    return b;
}

hipError_t launch_gemm_f16_prefill(const uint16_t* W, int R, int C, const float* X, int M, float* Y, void* workspace, hipStream_t st) {
    if (C % 32 != 0 || R % 4 != 0) return hipErrorInvalidValue;
    for (int m0 = 0; m0 < M; m0 += 128) {                       // token chunks of <= 128
        const int mc = (M - m0) < 128 ? (M - m0) : 128;
        const int MT = (mc + 31) / 32, Mpad = MT * 32;
        _Float16* hi = reinterpret_cast<_Float16*>(workspace);
        _Float16* lo = hi + (size_t)Mpad * C;
        const size_t n = (size_t)Mpad * C;
        const unsigned sgrid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(split_hi_lo_kernel, dim3(sgrid), dim3(256), 0, st, X + (size_t)m0 * C, mc, Mpad, C, hi, lo);
        float* Yc = Y + (size_t)m0 * R;
        if (C % kKC == 0) {
            const int ks = prefill_splits(R, C), nchunks = C / kKC, cps = (nchunks + ks - 1) / ks;
            float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ((size_t)Mpad * C * 2 * 2 + 255) / 256 * 256);
            float* dst = ks == 1 ? Yc : part;
            const dim3 grid((R + 127) / 128, ks);
            switch (MT) {
                case 1: hipLaunchKernelGGL(gemm_prefill_v2_kernel<1>, grid, dim3(256), 0, st, W, R, C, hi, lo, mc, dst, cps); break;
                case 2: hipLaunchKernelGGL(gemm_prefill_v2_kernel<2>, grid, dim3(256), 0, st, W, R, C, hi, lo, mc, dst, cps); break;
                case 3: hipLaunchKernelGGL(gemm_prefill_v2_kernel<3>, grid, dim3(256), 0, st, W, R, C, hi, lo, mc, dst, cps); break;
                default: hipLaunchKernelGGL(gemm_prefill_v2_kernel<4>, grid, dim3(256), 0, st, W, R, C, hi, lo, mc, dst, cps); break;
            }
            if (ks > 1) {
                const size_t n4 = (size_t)mc * R / 4;
                hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, part, ks, n4, Yc);
            }
        } else {                                                // odd K: v1 kernel (one wave per workgroup)
            const int grid = (R + 31) / 32;
            switch (MT) {
                case 1: hipLaunchKernelGGL(gemm_prefill_kernel<1>, dim3(grid), dim3(64), 0, st, W, R, C, hi, lo, mc, Yc); break;
                case 2: hipLaunchKernelGGL(gemm_prefill_kernel<2>, dim3(grid), dim3(64), 0, st, W, R, C, hi, lo, mc, Yc); break;
                case 3: hipLaunchKernelGGL(gemm_prefill_kernel<3>, dim3(grid), dim3(64), 0, st, W, R, C, hi, lo, mc, Yc); break;
                default: hipLaunchKernelGGL(gemm_prefill_kernel<4>, dim3(grid), dim3(64), 0, st, W, R, C, hi, lo, mc, Yc); break;
            }
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace thk
