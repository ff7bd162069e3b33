// thk_engine.hip — persistent loader/consumer decode engine for gfx950.
//
// ONE launch runs every layer of a decode step (reference: the per-token re-dispatch of th_eval_gpu,
// th-llama.cpp:464-660, and the blocking drain :686-706).  One 256-thread workgroup per CU:
//   wave 0      LOADER   streams this CU's weight rows of every op, in op order, through a ring of 16 KiB LDS slots with
//                        16 x 1 KiB `global_load_lds ... nt` per fill, at most one older fill outstanding (vmcnt(n)); it
//                        never waits for activations, only for ring space, so HBM keeps streaming across op boundaries
//                        (measured 6.9-7.1 TB/s chip-wide, profiles/r02_engine_probe3.txt).
//   waves 1-3   CONSUMERS  dot the slots (ds_read_b128) with the activation vector held in LDS, run the fused epilogue
//                        (RoPE + KV append | +residual | SwiGLU | logits + arg-max) and publish every output element as an
//                        8-byte {value, tag} granule with one agent-scope store.  The next op's input is GATHERED from
//                        the granules by the three consumer waves of every CU (each sweeps a third, re-reads until every
//                        tag matches, applies RMSNorm*gain where the op needs it) into LDS.
// Attention runs on the consumer waves with plain loads of the f32 cache ((head, split) per CU, first batch prefetched
// before q arrives, split combine by the split-0 CU of each head), while the loader is already filling the ring with wo.
// Protocol = guide recipe R2 (MI355X_MICROARCH.md "Persistent kernels", cdna_hip_programming.md G16): tag = (epoch, op),
// epoch is a device word bumped by the kernel that follows the engine in the stream, so nothing is zeroed per launch and a
// captured graph replays correctly.  Every wait is bounded; a time-out writes an error word and the workgroup leaves.
// LDS control words are touched only with ds_* instructions written as asm: a generic-pointer access compiles to
// flat_load + s_waitcnt vmcnt(0), which drains the loader's DMA queue (what made round 1's probe read 3.7 TB/s).
//
// Arithmetic is the launch path's (thk_kernels.hip): f16 weights decoded by hardware, f32 FMA, f32 activations;
// summation order inside a row differs (16-byte pieces per lane, then a DPP tree), within the 1e-3 logit tolerance.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "thk_kernels.hpp"
#include "thk_device.hpp"

namespace thk {

// One build (round 3): waiting consumer waves park one landed ring slot each in registers, which extends the loader's run-ahead
// by three slots.  (Round 2 also shipped the variant without parking; it was never faster and is gone.)
namespace eng {
#include "thk_engine_body.inc"
}

size_t engine_lds_bytes(int NS, int v0_bytes, int v1_bytes) { return (size_t)NS * kEngSlotBytes + (size_t)v0_bytes + (size_t)v1_bytes + eng::SC_BYTES + eng::CT_BYTES; }

hipError_t launch_engine(const EngArgs& a, int n_cu, hipStream_t st) {
    const size_t lds = engine_lds_bytes(a.NS, a.v0_bytes, a.v1_bytes);
    static size_t attr_set[kMaxDevices] = {};
    auto kn = eng::engine_kernel;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    if (lds > attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set[dev] = lds;
    }
    hipLaunchKernelGGL(kn, dim3(n_cu), dim3(256), lds, st, a, a.ops, a.st, a.epoch);
    return hipGetLastError();
}

}  // namespace thk
