// thk_peer.hip — the pipeline hand-off WITHOUT a communication library (SURVEY.md §8e fallback): every stage owns a
// "mailbox" in its own HBM, exported to the previous stage with hipIpcGetMemHandle (dmabuf IPC; HSA_ENABLE_IPC_MODE_LEGACY=0);
// the producing stage's stream runs a one-workgroup kernel that stores the f32 hidden state (or the 4-byte token) straight
// into the consumer's mailbox over xGMI and then raises a sequence-numbered flag next to it; the consuming stage's stream
// runs a kernel that waits for the flag and copies the payload into the model's hidden_in / token slot.
//
//   * One ncclSend/ncclRecv pair costs two kernel launches plus RCCL's proxy bookkeeping per 16 KB; here the hand-off is the
//     16 KB store itself + one 8-byte flag.  It is an OPT-IN alternative (bench.py --transport peer), not the default: on this
//     project's hardware access it could only be exercised with two processes on ONE GPU (tests/test_gpu_pipeline.py), never
//     across xGMI.
//   * Ordering: payload with system-scope stores, __threadfence_system(), explicit s_waitcnt vmcnt(0) (the compiler may drop
//     the wait between a release fence and a following store, MI355X_MICROARCH.md), then the flag with a system-scope release
//     store.  The consumer polls the flag with system-scope acquire loads (one lane, s_sleep) and reads the payload with
//     system-scope loads, so no cached copy on either side is ever trusted.
//   * Flags are monotonically increasing per (sequence, kind); sender and receiver keep their own device-resident counters, so
//     the kernels take constant arguments (a captured graph would replay them) and nothing is ever reset.
//   * Flow control is the ring itself: a stage produces item i + S for a sequence only after the token/hidden of item i has
//     travelled through every other stage, so a mailbox slot is never overwritten before it was read (S = N sequences in flight).
//   * Every wait is bounded (~2 s on the 100 MHz counter): a missing peer raises an error word (thk_peer_check) instead of
//     hanging the GPU.  A time-out is STICKY: the flag sequence of that (sequence, kind) is out of step from then on, so every
//     later wait returns at once without touching its destination, thk_peer_check keeps reporting the failure and
//     thk_peer_send / thk_peer_recv refuse - the peer has to be destroyed and recreated (round 4; round 3 cleared the word and let
//     later hand-offs pair with stale flags).
//   * The mailbox is FINE-GRAINED device memory (hipExtMallocWithFlags: uncached, else fine-grained): another GPU writes it while
//     this GPU's kernel polls it, and HIP only guarantees cross-agent visibility of ordinary (coarse-grained) allocations at
//     dispatch boundaries.  If neither flavour can be allocated AND exported over IPC the mailbox falls back to hipMalloc and
//     thk_peer_memory_kind says so (then only same-GPU rings are safe - what tests/test_gpu_pipeline.py runs).
//   * Bulk payloads (round 6: the [n_tokens, E] rows a stage's prompt pass hands to the next stage, thk_model_prefill_stage) have a slot of
//     their own per sequence (n_ctx * E * 4 bytes) and a flag of their own: a wide copy kernel stores the rows into the next stage's slot,
//     a one-thread kernel BEHIND it on the stream raises the flag (the copy kernel's threads fence their system-scope stores before it ends);
//     the consumer's one-workgroup wait kernel is followed by a wide copy out of its own mailbox.  Flow control: a sequence's bulk slot is
//     written once per prompt; the caller synchronises + fences across ranks before the same sequence's next prompt (PipelineDriver.prefill).
// Reference: none - the reference is single-device.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include "../../include/thk.h"

struct thk_ctx;
extern "C" void* thk_ctx_stream(thk_ctx* ctx);
namespace thk { int ctx_fail(thk_ctx* ctx, int code, const char* msg); int ctx_device(thk_ctx* ctx); }

namespace {
typedef unsigned long long u64;
constexpr int kLine = 16;                         // u64 per 128-byte line
constexpr u64 kTimeoutTicks = 200000000ull;       // 2 s of the 100 MHz s_memrealtime counter

// producer: src (this GPU) -> dst (next stage's mailbox slot), then flag = ++sent
// n_words == 0: the payload is ONE 32-bit word (the token id, which sits at a 4-byte offset inside the sequence state)
__global__ __launch_bounds__(256) void peer_push_kernel(const u64* __restrict__ src, int n_words, u64* dst, u64* flag, u64* sent) {
    for (int i = threadIdx.x; i < n_words; i += 256) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (n_words == 0 && threadIdx.x == 0)
        __hip_atomic_store(reinterpret_cast<unsigned*>(dst), *reinterpret_cast<const unsigned*>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const u64 v = *sent + 1;
        *sent = v;
        __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// consumer: wait for flag >= ++received (bounded), then slot (own mailbox) -> dst (hidden_in / token)
__global__ __launch_bounds__(256) void peer_wait_kernel(const u64* slot, int n_words, u64* __restrict__ dst, const u64* flag, u64* received, unsigned* err) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
        const u64 want = *received + 1;
        const u64 t0 = __builtin_amdgcn_s_memrealtime();
        int good = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;     // an earlier time-out: the chain is broken, do not wait again
        while (good && __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
            __builtin_amdgcn_s_sleep(16);
            if (__builtin_amdgcn_s_memrealtime() - t0 > kTimeoutTicks) { good = 0; break; }
        }
        if (good) *received = want;
        else __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok = good;
    }
    __syncthreads();
    if (!ok) return;
    for (int i = threadIdx.x; i < n_words; i += 256) dst[i] = __hip_atomic_load(slot + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (n_words == 0 && threadIdx.x == 0)
        *reinterpret_cast<unsigned*>(dst) = __hip_atomic_load(reinterpret_cast<const unsigned*>(slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// bulk payload, grid-stride over 8-byte words; the flag travels in a kernel of its own behind this one.  The mailbox side of the copy uses the same
// system-scope accesses as the small payloads above (PUSH: stores into the NEXT stage's mailbox; !PUSH: loads from this stage's own mailbox, which
// another GPU wrote): no cached copy on either side is trusted, whatever flavour of fine-grained memory the mailbox got
// err (!PUSH): the error word of the wait kernel ahead of this one on the stream - after a time-out nothing arrived, so nothing is copied (the destination keeps
// what it held; thk_peer_check reports the failure)
template <bool PUSH>
__global__ __launch_bounds__(256) void peer_bulk_copy_kernel(const u64* __restrict__ src, u64* __restrict__ dst, size_t n8, const unsigned* err) {
    if (!PUSH && err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        if (PUSH) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (PUSH) { __threadfence_system(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }      // every thread's stores are on their way out before the kernel ends; the flag kernel follows on the stream
}
__global__ void peer_bulk_flag_kernel(u64* flag, u64* sent) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        __threadfence_system();
        const u64 v = *sent + 1;
        *sent = v;
        __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
}  // namespace

// Mailbox of one stage (exported): per sequence a hidden slot of E/2 words and a token slot of one word (padded to a line),
// then one flag line per (sequence, kind).  Counters (sent / received) and the error word are private.
struct thk_peer {
    thk_ctx* ctx = nullptr;
    thk_model* model = nullptr;
    int n_seq = 1, hidden_words = 0;
    size_t bulk_words = 0;       // per sequence: n_ctx * E / 2 (the rows of a whole prompt), line-rounded
    u64* box = nullptr;          // own mailbox (device)
    int box_kind = 0;            // THK_PEER_MEM_*: how it was allocated
    bool failed = false;         // a hand-off wait timed out (sticky)
    size_t box_words = 0;
    u64* next_box = nullptr;     // the next stage's mailbox as mapped into this process
    bool next_is_ipc = false;
    u64* counters = nullptr;     // [n_seq][2 kinds][sent, received] + error word, private
    unsigned* err = nullptr;
    size_t hidden_off(int s) const { return (size_t)s * hidden_words; }
    size_t token_off(int s) const { return (size_t)n_seq * hidden_words + (size_t)s * kLine; }
    size_t flag_off(int s, int kind) const { return (size_t)n_seq * hidden_words + (size_t)n_seq * kLine + ((size_t)s * 3 + kind) * kLine; }
    size_t bulk_off(int s) const { return flag_off(n_seq, 0) + (size_t)s * bulk_words; }      // behind the flags; 128-byte aligned
};
#define PEERCHK(p, call)                                                                                 \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) { char b_[256]; snprintf(b_, sizeof b_, "%s failed: %s", #call, hipGetErrorString(e_)); return thk::ctx_fail((p), THK_ERR_HIP, b_); } \
    } while (0)

extern "C" int thk_peer_create(thk_ctx* ctx, thk_model* stage, int32_t n_seq, thk_peer** out) {
    if (!ctx || !stage || !out || n_seq < 1) return THK_ERR_INVALID;
    *out = nullptr;
    const int E = thk_model_n_embd(stage);
    if (E <= 0 || (E & 1) || !thk_model_hidden_in(stage, n_seq - 1)) return thk::ctx_fail(ctx, THK_ERR_INVALID, "thk_peer_create: the stage must be finalized with at least n_seq sequences");
    PEERCHK(ctx, hipSetDevice(thk::ctx_device(ctx)));
    thk_peer* p = new thk_peer();
    p->ctx = ctx; p->model = stage; p->n_seq = n_seq; p->hidden_words = (E / 2 + kLine - 1) / kLine * kLine;
    p->bulk_words = ((size_t)thk_model_n_ctx(stage) * E / 2 + kLine - 1) / kLine * kLine;
    p->box_words = p->bulk_off(n_seq);
    hipStream_t st = (hipStream_t)thk_ctx_stream(ctx);
    // its own allocation (the IPC handle covers exactly the mailbox), fine-grained if the runtime can allocate AND export that
    hipError_t e = hipErrorUnknown;
    const struct { unsigned flag; int kind; } tries[] = {{hipDeviceMallocUncached, THK_PEER_MEM_UNCACHED}, {hipDeviceMallocFinegrained, THK_PEER_MEM_FINEGRAINED}};
    for (const auto& t : tries) {
        void* q = nullptr;
        if (hipExtMallocWithFlags(&q, p->box_words * 8, t.flag) != hipSuccess) { (void)hipGetLastError(); continue; }
        hipIpcMemHandle_t h;
        if (hipIpcGetMemHandle(&h, q) != hipSuccess) { (void)hipGetLastError(); hipFree(q); continue; }
        p->box = (u64*)q; p->box_kind = t.kind; e = hipSuccess;
        break;
    }
    if (!p->box) { e = hipMalloc((void**)&p->box, p->box_words * 8); p->box_kind = THK_PEER_MEM_COARSE; }
    if (e == hipSuccess) e = hipMalloc((void**)&p->counters, ((size_t)n_seq * 6 + 2) * 8);
    if (e == hipSuccess) e = hipMemsetAsync(p->box, 0, p->box_words * 8, st);
    if (e == hipSuccess) e = hipMemsetAsync(p->counters, 0, ((size_t)n_seq * 6 + 2) * 8, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { hipFree(p->box); hipFree(p->counters); delete p; return thk::ctx_fail(ctx, THK_ERR_HIP, "thk_peer_create: mailbox allocation failed"); }
    p->err = reinterpret_cast<unsigned*>(p->counters + (size_t)n_seq * 6);
    *out = p;
    return THK_OK;
}
extern "C" int thk_peer_memory_kind(const thk_peer* p) { return p ? p->box_kind : THK_ERR_INVALID; }
extern "C" int thk_peer_export(thk_peer* p, void* handle_out64) {
    if (!p || !handle_out64) return THK_ERR_INVALID;
    static_assert(sizeof(hipIpcMemHandle_t) == THK_PEER_HANDLE_BYTES, "hipIpcMemHandle_t size");
    hipIpcMemHandle_t h;
    PEERCHK(p->ctx, hipSetDevice(thk::ctx_device(p->ctx)));
    PEERCHK(p->ctx, hipIpcGetMemHandle(&h, p->box));
    memcpy(handle_out64, &h, sizeof h);
    return THK_OK;
}
// next_handle64 == NULL: a single-stage ring (the stage is its own successor: same process, no IPC)
extern "C" int thk_peer_connect(thk_peer* p, const void* next_handle64) {
    if (!p) return THK_ERR_INVALID;
    PEERCHK(p->ctx, hipSetDevice(thk::ctx_device(p->ctx)));
    if (!next_handle64) { p->next_box = p->box; p->next_is_ipc = false; return THK_OK; }
    hipIpcMemHandle_t h;
    memcpy(&h, next_handle64, sizeof h);
    void* ptr = nullptr;
    PEERCHK(p->ctx, hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
    p->next_box = (u64*)ptr; p->next_is_ipc = true;
    return THK_OK;
}
static int peer_io(thk_peer* p, int32_t seq, int kind, bool send) {
    if (!p || seq < 0 || seq >= p->n_seq || (kind != THK_PEER_HIDDEN && kind != THK_PEER_TOKEN)) return THK_ERR_INVALID;
    if (!p->next_box) return thk::ctx_fail(p->ctx, THK_ERR_STATE, "thk_peer: thk_peer_connect first");
    if (p->failed) return thk::ctx_fail(p->ctx, THK_ERR_STATE, "thk_peer: a hand-off timed out earlier; the flag sequence is out of step - destroy this peer and create a new one");
    hipStream_t st = (hipStream_t)thk_ctx_stream(p->ctx);
    const int words = kind == THK_PEER_HIDDEN ? thk_model_n_embd(p->model) / 2 : 0;
    const size_t off = kind == THK_PEER_HIDDEN ? p->hidden_off(seq) : p->token_off(seq);
    u64* cnt = p->counters + ((size_t)seq * 3 + kind) * 2;
    if (send) {
        const void* src = kind == THK_PEER_HIDDEN ? thk_model_hidden_out(p->model, seq) : thk_model_token_dev(p->model, seq);
        hipLaunchKernelGGL(peer_push_kernel, dim3(1), dim3(256), 0, st, (const u64*)src, words, p->next_box + off, p->next_box + p->flag_off(seq, kind), cnt);
    } else {
        void* dst = kind == THK_PEER_HIDDEN ? thk_model_hidden_in(p->model, seq) : thk_model_token_dev(p->model, seq);
        hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(256), 0, st, (const u64*)(p->box + off), words, (u64*)dst, (const u64*)(p->box + p->flag_off(seq, kind)), cnt + 1, p->err);
    }
    PEERCHK(p->ctx, hipGetLastError());
    return THK_OK;
}
// bulk: bytes of a caller-owned device buffer (the [n_tokens, E] rows of thk_model_prefill_stage) through the sequence's bulk slot
static int peer_bulk(thk_peer* p, int32_t seq, void* buf, size_t bytes, bool send) {
    if (!p || !buf || seq < 0 || seq >= p->n_seq) return THK_ERR_INVALID;
    if (bytes == 0 || (bytes & 15) || bytes > p->bulk_words * 8 || ((uintptr_t)buf & 15)) return thk::ctx_fail(p->ctx, THK_ERR_INVALID, "thk_peer bulk: bytes must be a multiple of 16, at most n_ctx * n_embd * 4, the buffer 16-byte aligned");
    if (!p->next_box) return thk::ctx_fail(p->ctx, THK_ERR_STATE, "thk_peer: thk_peer_connect first");
    if (p->failed) return thk::ctx_fail(p->ctx, THK_ERR_STATE, "thk_peer: a hand-off timed out earlier; the flag sequence is out of step - destroy this peer and create a new one");
    PEERCHK(p->ctx, hipSetDevice(thk::ctx_device(p->ctx)));
    hipStream_t st = (hipStream_t)thk_ctx_stream(p->ctx);
    u64* cnt = p->counters + ((size_t)seq * 3 + THK_PEER_BULK) * 2;
    const size_t n8 = bytes / 8;
    const int grid = (int)std::min<size_t>(1024, (n8 + 255) / 256);
    if (send) {
        hipLaunchKernelGGL(peer_bulk_copy_kernel<true>, dim3(grid), dim3(256), 0, st, (const u64*)buf, p->next_box + p->bulk_off(seq), n8, (const unsigned*)nullptr);
        hipLaunchKernelGGL(peer_bulk_flag_kernel, dim3(1), dim3(64), 0, st, p->next_box + p->flag_off(seq, THK_PEER_BULK), cnt);
    } else {
        hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(256), 0, st, (const u64*)nullptr, -1, (u64*)nullptr, (const u64*)(p->box + p->flag_off(seq, THK_PEER_BULK)), cnt + 1, p->err);
        hipLaunchKernelGGL(peer_bulk_copy_kernel<false>, dim3(grid), dim3(256), 0, st, (const u64*)(p->box + p->bulk_off(seq)), (u64*)buf, n8, (const unsigned*)p->err);
    }
    PEERCHK(p->ctx, hipGetLastError());
    return THK_OK;
}
extern "C" int thk_peer_send_bulk(thk_peer* p, int32_t seq, const void* src_dev, size_t bytes) { return peer_bulk(p, seq, const_cast<void*>(src_dev), bytes, true); }
extern "C" int thk_peer_recv_bulk(thk_peer* p, int32_t seq, void* dst_dev, size_t bytes) { return peer_bulk(p, seq, dst_dev, bytes, false); }
extern "C" int thk_peer_send(thk_peer* p, int32_t seq, int kind) { return peer_io(p, seq, kind, true); }
extern "C" int thk_peer_recv(thk_peer* p, int32_t seq, int kind) { return peer_io(p, seq, kind, false); }
// THK_ERR_STATE when a bounded wait gave up since the last check (the stream is synchronized first)
extern "C" int thk_peer_check(thk_peer* p) {
    if (!p) return THK_ERR_INVALID;
    hipStream_t st = (hipStream_t)thk_ctx_stream(p->ctx);
    unsigned e = 0;
    PEERCHK(p->ctx, hipMemcpyAsync(&e, p->err, 4, hipMemcpyDeviceToHost, st));
    PEERCHK(p->ctx, hipStreamSynchronize(st));
    if (e || p->failed) {
        p->failed = true;          // sticky: the device word stays set, so waits already enqueued return at once
        return thk::ctx_fail(p->ctx, THK_ERR_STATE, "thk_peer: a hand-off wait timed out (the previous stage never delivered); the peer is unusable from here on");
    }
    return THK_OK;
}
extern "C" int thk_peer_destroy(thk_peer* p) {
    if (!p) return THK_OK;
    hipSetDevice(thk::ctx_device(p->ctx));
    hipStreamSynchronize((hipStream_t)thk_ctx_stream(p->ctx));
    if (p->next_is_ipc && p->next_box) hipIpcCloseMemHandle(p->next_box);
    hipFree(p->box); hipFree(p->counters);
    delete p;
    return THK_OK;
}
