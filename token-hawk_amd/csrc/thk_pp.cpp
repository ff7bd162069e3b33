// thk_pp.cpp — pipeline-parallel point-to-point hand-off over RCCL/xGMI (config C4).
//
// The only inter-stage traffic of the layer pipeline is the f32 hidden state (E*4 bytes) from
// stage r to r+1 and the 4-byte greedy token from the last stage back to stage 0:
// ncclSend/ncclRecv on the context's stream, grouped so a ring step (one send + one receive per
// rank) cannot dead-lock.  No collective is used anywhere in the data path.
//
// librccl is bound lazily with dlopen the first time a thk_pp_* entry point is called, so libthk
// keeps loading on hosts without RCCL and never competes with a copy already loaded by the caller
// (e.g. the one bundled with PyTorch: dlopen by soname returns the resident library).
#include "../../include/thk.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>
#include <string>

struct thk_ctx;   // defined in thk_internal.hpp
extern "C" void* thk_ctx_stream(thk_ctx* ctx);
namespace thk { int ctx_fail(thk_ctx* ctx, int code, const char* msg); int ctx_device(thk_ctx* ctx); }

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
Rccl& rccl() {
    static Rccl r;
    if (r.handle || !r.error.empty()) return r;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) { const char* de = dlerror(); r.error = std::string("cannot load librccl: ") + (de ? de : "unknown dlopen error"); return r; }
#define BIND(field, sym)                                                                 \
    *(void**)(&r.field) = dlsym(r.handle, sym);                                          \
    if (!r.field) { r.error = std::string("librccl lacks ") + sym; r.handle = nullptr; return r; }
    BIND(GetUniqueId, "ncclGetUniqueId") BIND(CommInitRank, "ncclCommInitRank") BIND(CommDestroy, "ncclCommDestroy")
    BIND(Send, "ncclSend") BIND(Recv, "ncclRecv") BIND(GroupStart, "ncclGroupStart") BIND(GroupEnd, "ncclGroupEnd")
    BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
    return r;
}
}  // namespace

struct thk_pp {
    thk_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int n_ranks = 1, rank = 0;
};

#define RCCLCHK(ctx, call)                                                                           \
    do {                                                                                             \
        ncclResult_t r_ = (call);                                                                    \
        if (r_ != ncclSuccess) { char b_[256]; snprintf(b_, sizeof b_, "%s failed: %s", #call, rccl().GetErrorString(r_)); return thk::ctx_fail((ctx), THK_ERR_RCCL, b_); } \
    } while (0)

extern "C" int thk_pp_get_unique_id(void* out128) {
    if (!out128) return THK_ERR_INVALID;
    Rccl& r = rccl();
    if (!r.handle) return THK_ERR_RCCL;
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) return THK_ERR_RCCL;
    static_assert(sizeof(id) == THK_PP_UNIQUE_ID_BYTES, "ncclUniqueId size");
    memcpy(out128, &id, sizeof id);
    return THK_OK;
}

extern "C" int thk_pp_create(thk_ctx* ctx, int n_ranks, int rank, const void* unique_id128, thk_pp** out) {
    if (!ctx || !out || !unique_id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return THK_ERR_INVALID;
    *out = nullptr;
    Rccl& r = rccl();
    if (!r.handle) return thk::ctx_fail(ctx, THK_ERR_RCCL, r.error.c_str());
    if (hipSetDevice(thk::ctx_device(ctx)) != hipSuccess) return thk::ctx_fail(ctx, THK_ERR_HIP, "hipSetDevice failed");
    thk_pp* pp = new thk_pp();
    pp->ctx = ctx; pp->n_ranks = n_ranks; pp->rank = rank;
    ncclUniqueId id;
    memcpy(&id, unique_id128, sizeof id);
    ncclResult_t rc = r.CommInitRank(&pp->comm, n_ranks, id, rank);
    if (rc != ncclSuccess) { delete pp; char b[256]; snprintf(b, sizeof b, "ncclCommInitRank failed: %s", r.GetErrorString(rc)); return thk::ctx_fail(ctx, THK_ERR_RCCL, b); }
    *out = pp;
    return THK_OK;
}
extern "C" int thk_pp_destroy(thk_pp* pp) {
    if (!pp) return THK_OK;
    hipStreamSynchronize((hipStream_t)thk_ctx_stream(pp->ctx));
    if (pp->comm) rccl().CommDestroy(pp->comm);
    delete pp;
    return THK_OK;
}
extern "C" int thk_pp_rank(const thk_pp* pp) { return pp ? pp->rank : -1; }
extern "C" int thk_pp_size(const thk_pp* pp) { return pp ? pp->n_ranks : 0; }
extern "C" int thk_pp_group_begin(thk_pp* pp) {
    if (!pp) return THK_ERR_INVALID;
    RCCLCHK(pp->ctx, rccl().GroupStart());
    return THK_OK;
}
extern "C" int thk_pp_group_end(thk_pp* pp) {
    if (!pp) return THK_ERR_INVALID;
    RCCLCHK(pp->ctx, rccl().GroupEnd());
    return THK_OK;
}
extern "C" int thk_pp_send(thk_pp* pp, const void* dev_buf, size_t bytes, int peer) {
    if (!pp || !dev_buf || peer < 0 || peer >= pp->n_ranks) return THK_ERR_INVALID;
    RCCLCHK(pp->ctx, rccl().Send(dev_buf, bytes, ncclChar, peer, pp->comm, (hipStream_t)thk_ctx_stream(pp->ctx)));
    return THK_OK;
}
extern "C" int thk_pp_recv(thk_pp* pp, void* dev_buf, size_t bytes, int peer) {
    if (!pp || !dev_buf || peer < 0 || peer >= pp->n_ranks) return THK_ERR_INVALID;
    RCCLCHK(pp->ctx, rccl().Recv(dev_buf, bytes, ncclChar, peer, pp->comm, (hipStream_t)thk_ctx_stream(pp->ctx)));
    return THK_OK;
}
extern "C" int thk_pp_send_hidden(thk_pp* pp, thk_model* m, int32_t seq, int peer) {
    void* p = thk_model_hidden_out(m, seq);
    if (!pp || !p) return THK_ERR_INVALID;
    return thk_pp_send(pp, p, (size_t)thk_model_n_embd(m) * 4, peer);
}
extern "C" int thk_pp_recv_hidden(thk_pp* pp, thk_model* m, int32_t seq, int peer) {
    void* p = thk_model_hidden_in(m, seq);
    if (!pp || !p) return THK_ERR_INVALID;
    return thk_pp_recv(pp, p, (size_t)thk_model_n_embd(m) * 4, peer);
}
extern "C" int thk_pp_send_token(thk_pp* pp, thk_model* m, int32_t seq, int peer) {
    void* p = thk_model_token_dev(m, seq);
    if (!pp || !p) return THK_ERR_INVALID;
    return thk_pp_send(pp, p, 4, peer);
}
extern "C" int thk_pp_recv_token(thk_pp* pp, thk_model* m, int32_t seq, int peer) {
    void* p = thk_model_token_dev(m, seq);
    if (!pp || !p) return THK_ERR_INVALID;
    return thk_pp_recv(pp, p, 4, peer);
}
