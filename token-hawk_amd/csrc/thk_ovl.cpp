// thk_ovl.cpp — overlapped dispatch of the decode step (tunable "overlap_dispatch").
//
// A HIP stream (and a hipGraph replay of it) sets the AQL barrier bit on every kernel packet: a launch is not even placed on the
// CUs before its predecessor's last wave has retired, and on MI355X that boundary costs ~1.3 us of idle chip plus the ~1.8 us
// ramp of the next launch, 162 times per LLaMA-7B token.  Here the same 162 launches are written as AQL packets
// to a user-mode queue of our own, and chosen packets (tunable overlap_keep_barrier) go WITHOUT the barrier bit: the command
// processor may place such a kernel's workgroups while its predecessor still runs, every wave requests its first batch of
// weights (they depend on nothing), and the dependency itself is enforced inside the kernels (thk_device.hpp: sharded arrival
// counters, agent-coherent accesses for everything that crosses a launch; kernels in thk_ovl_kernels.hip -> libthk_ovl.hsaco,
// loaded with the HSA runtime).  A step's first packet always keeps the bit, so a step starts after the previous one has
// completely finished (its last launch zeroes the counters).
// MEASURED OUTCOME (profiles/r03_overlap_ab.txt): equal results, not faster than hipGraph replays - the command
// processor starts a barrier-free successor only in its predecessor's last microseconds and the protocol costs what that buys.
// The option stays for experiments; the default mask overlaps the two boundaries that gain (wo -> w1|w3, w2 -> qkv).
//
// The queue is ordered against the ctx stream with two pairs of tiny kernels: the stream writes a ticket that the batch's first
// packet waits for, the batch's last packet bumps a counter that a kernel enqueued on the stream waits for.  So
// thk_model_decode_step(s) keep their stream-ordered meaning: whatever was enqueued before is visible to the batch, whatever is
// enqueued after sees its results, hipStreamSynchronize / hipDeviceSynchronize cover it.
//
// Kernel arguments are constant per (sequence, launch) - positions, tokens and counters live in device memory - and are written
// once, into DEVICE memory (arguments in host memory make every wave's first scalar loads cross PCIe: 2x slower, measured with
// tools/probes/aql_overlap_probe).  Probe result that motivated this: profiles/r03_aql_overlap_probe.txt.
#include "thk_internal.hpp"

#include <dlfcn.h>
#include <stddef.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

namespace {

struct OvlKernel { uint64_t object = 0; uint32_t kernarg = 0, lds_static = 0, scratch = 0; };

struct OvlQueue {
    bool hsa_up = false;
    hsa_agent_t agent{};
    hsa_code_object_reader_t reader{}; bool have_reader = false;
    hsa_executable_t exe{}; bool have_exe = false;
    hsa_queue_t* q = nullptr;
    std::vector<char> blob;
    std::map<std::string, OvlKernel> kernels;
    unsigned* words = nullptr;        // device: [0] ticket (written by the ctx stream), [32] completed batches, [64] error word
    void* gate_args = nullptr;        // device: argument block of the batch_begin / batch_end kernels
    OvlKernel k_begin, k_end;
    unsigned batch = 0;               // batches submitted
};

struct OvlProgram {
    std::vector<hsa_kernel_dispatch_packet_t> packets;   // one decode step, headers kept aside
    std::vector<uint32_t> headers;                       // header | setup << 16
    void* kargs = nullptr;                               // device
};

const char* hsa_err(hsa_status_t s) { const char* m = nullptr; hsa_status_string(s, &m); return m ? m : "?"; }
#define HSACHK(ctx, call)                                                                                  \
    do {                                                                                                   \
        hsa_status_t s_ = (call);                                                                          \
        if (s_ != HSA_STATUS_SUCCESS && s_ != HSA_STATUS_INFO_BREAK) return fail((ctx), THK_ERR_HIP, "%s failed: %s (%s:%d)", #call, hsa_err(s_), __FILE__, __LINE__); \
    } while (0)

uint32_t make_header(bool barrier, int acquire, int release) {
    const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                                       (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
    const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
    return (uint32_t)header | ((uint32_t)setup << 16);
}
hsa_kernel_dispatch_packet_t make_packet(const OvlKernel& k, int grid, int block, int lds_dynamic, void* kernarg) {
    hsa_kernel_dispatch_packet_t p{};
    p.workgroup_size_x = (uint16_t)block; p.workgroup_size_y = 1; p.workgroup_size_z = 1;
    p.grid_size_x = (uint32_t)grid * (uint32_t)block; p.grid_size_y = 1; p.grid_size_z = 1;
    p.private_segment_size = k.scratch; p.group_segment_size = k.lds_static + (uint32_t)lds_dynamic;
    p.kernel_object = k.object; p.kernarg_address = kernarg; p.completion_signal = hsa_signal_t{0};
    return p;
}

struct AgentSearch { uint32_t domain, bdf; int ordinal, seen; hsa_agent_t by_bdf, by_ordinal; bool have_bdf, have_ordinal; };
hsa_status_t agent_cb(hsa_agent_t a, void* data) {
    AgentSearch* s = static_cast<AgentSearch*>(data);
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    uint32_t bdf = 0, dom = 0;
    const bool ok = hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) == HSA_STATUS_SUCCESS &&
                    hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom) == HSA_STATUS_SUCCESS;
    if (ok && bdf == s->bdf && dom == s->domain && !s->have_bdf) { s->by_bdf = a; s->have_bdf = true; }
    if (s->seen == s->ordinal && !s->have_ordinal) { s->by_ordinal = a; s->have_ordinal = true; }
    s->seen += 1;
    return HSA_STATUS_SUCCESS;
}

std::string hsaco_path() {
    if (const char* e = getenv("THK_OVL_HSACO")) return e;
    Dl_info info{};
    if (dladdr((const void*)&thk_model_create, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        const size_t slash = p.rfind('/');
        return (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + (trace_compiled() ? "/libthk_ovl_trace.hsaco" : "/libthk_ovl.hsaco");
    }
    return "libthk_ovl.hsaco";
}

int load_kernel(thk_ctx* ctx, OvlQueue* Q, const char* name, OvlKernel* out) {
    auto it = Q->kernels.find(name);
    if (it != Q->kernels.end()) { *out = it->second; return THK_OK; }
    hsa_executable_symbol_t sym;
    const std::string kd = std::string(name) + ".kd";
    if (hsa_executable_get_symbol_by_name(Q->exe, kd.c_str(), &Q->agent, &sym) != HSA_STATUS_SUCCESS)
        return fail(ctx, THK_ERR_NOTFOUND, "overlapped dispatch: kernel %s is not in libthk_ovl.hsaco (this launch geometry has no overlapped variant; see thk_ovl_kernels.hip)", name);
    OvlKernel k;
    HSACHK(ctx, hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    HSACHK(ctx, hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg));
    HSACHK(ctx, hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.lds_static));
    HSACHK(ctx, hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.scratch));
    Q->kernels[name] = k;
    *out = k;
    return THK_OK;
}

int queue_up(thk_ctx* ctx, OvlQueue** out) {
    if (ctx->ovl) { *out = static_cast<OvlQueue*>(ctx->ovl); return THK_OK; }
    OvlQueue* Q = new OvlQueue();
    ctx->ovl = Q;                                     // owned by the ctx from here on (ovl_destroy), also when the set-up fails half-way
    HSACHK(ctx, hsa_init());                          // reference-counted: the HIP runtime of this process holds the first reference
    Q->hsa_up = true;
    char bus[64] = {0};
    AgentSearch s{};
    s.ordinal = ctx->device;
    unsigned dom = 0, b = 0, d = 0, f = 0;
    if (hipDeviceGetPCIBusId(bus, sizeof bus, ctx->device) == hipSuccess && sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) == 4) { s.domain = dom; s.bdf = (b << 8) | (d << 3) | f; }
    else s.bdf = 0xFFFFFFFFu;
    HSACHK(ctx, hsa_iterate_agents(agent_cb, &s));
    if (s.have_bdf) Q->agent = s.by_bdf;
    else if (s.have_ordinal) Q->agent = s.by_ordinal;
    else return fail(ctx, THK_ERR_STATE, "overlapped dispatch: no HSA agent for HIP device %d (%s)", ctx->device, bus);
    // code object
    const std::string path = hsaco_path();
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) return fail(ctx, THK_ERR_NOTFOUND, "overlapped dispatch: cannot open %s (built by __graft_entry__.build_libthk next to libthk.so)", path.c_str());
    fseek(fp, 0, SEEK_END); const long sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    Q->blob.resize(sz > 0 ? sz : 0);
    const size_t got = sz > 0 ? fread(Q->blob.data(), 1, sz, fp) : 0;
    fclose(fp);
    if (sz <= 0 || got != (size_t)sz) return fail(ctx, THK_ERR_STATE, "overlapped dispatch: short read of %s", path.c_str());
    HSACHK(ctx, hsa_code_object_reader_create_from_memory(Q->blob.data(), Q->blob.size(), &Q->reader)); Q->have_reader = true;
    hsa_profile_t prof;
    HSACHK(ctx, hsa_agent_get_info(Q->agent, HSA_AGENT_INFO_PROFILE, &prof));
    HSACHK(ctx, hsa_executable_create_alt(prof, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &Q->exe)); Q->have_exe = true;
    HSACHK(ctx, hsa_executable_load_agent_code_object(Q->exe, Q->agent, Q->reader, nullptr, nullptr));
    HSACHK(ctx, hsa_executable_freeze(Q->exe, nullptr));
    uint32_t qmax = 0;
    HSACHK(ctx, hsa_agent_get_info(Q->agent, HSA_AGENT_INFO_QUEUE_MAX_SIZE, &qmax));
    uint32_t qsize = 16384;
    while (qsize > qmax && qsize > 256) qsize >>= 1;
    HSACHK(ctx, hsa_queue_create(Q->agent, qsize, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &Q->q));
    int rc = load_kernel(ctx, Q, "thk_ovl_batch_begin", &Q->k_begin);
    if (rc == THK_OK) rc = load_kernel(ctx, Q, "thk_ovl_batch_end", &Q->k_end);
    if (rc != THK_OK) return rc;
    HIPCHK(ctx, hipMalloc((void**)&Q->words, 128 * 4));
    HIPCHK(ctx, hipMemset(Q->words, 0, 128 * 4));
    HIPCHK(ctx, hipMalloc(&Q->gate_args, 512));
    HIPCHK(ctx, hipMemset(Q->gate_args, 0, 512));
    HIPCHK(ctx, hipMemcpy(Q->gate_args, &Q->words, sizeof(unsigned*), hipMemcpyHostToDevice));
    *out = Q;
    return THK_OK;
}

// write n packets (bodies + headers) behind each other; the doorbell is rung every 128 packets and at the end, so a batch longer
// than the ring drains while it is being written
int push_packets(thk_ctx* ctx, OvlQueue* Q, const hsa_kernel_dispatch_packet_t* pk, const uint32_t* hdr, size_t n, int* since_bell) {
    hsa_queue_t* q = Q->q;
    for (size_t i = 0; i < n; ++i) {
        const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
        uint64_t spins = 0;
        while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {
            if (++spins > 4000000000ull) return fail(ctx, THK_ERR_STATE, "overlapped dispatch: the queue does not drain");
        }
        hsa_kernel_dispatch_packet_t* dst = (hsa_kernel_dispatch_packet_t*)q->base_address + (idx & (q->size - 1));
        memcpy((char*)dst + 4, (const char*)&pk[i] + 4, sizeof *dst - 4);
        __atomic_store_n((uint32_t*)dst, hdr[i], __ATOMIC_RELEASE);
        if (++*since_bell >= 128) { hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx); *since_bell = 0; }
        if (i + 1 == n && *since_bell) { /* the caller rings at the end of the batch */ }
    }
    return THK_OK;
}

}  // namespace

__global__ void ovl_ticket_kernel(unsigned* words, unsigned k) {
    if (threadIdx.x == 0) __hip_atomic_store(words, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void ovl_join_kernel(unsigned* words, unsigned k, unsigned long long limit_ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((int)(__hip_atomic_load(words + 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - k) < 0) {
        __builtin_amdgcn_s_sleep(64);
        if (__builtin_amdgcn_s_memrealtime() - t0 > limit_ticks) {
            if (threadIdx.x == 0) __hip_atomic_store(words + 64, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
}

bool ovl_eligible(const thk_model* m, const char** why) {
    const char* w = nullptr;
    const int E = m->hp.n_embd;
    if (!(m->flags & THK_STAGE_EMBED) || !(m->flags & THK_STAGE_HEAD)) w = "a pipeline stage (the overlapped step starts at the embedding and ends with the greedy pick)";
    else if (m->engine) w = "the one-launch engine is on";
    else if (!m->fold_embed) w = "fold_embed = 0";
    else if (m->skip_kernel == 1 || m->skip_kernel == 6) w = "measure_skip_kernel drops the step's first or last mat-vec";
    else if (m->nsplit == 1) w = "attn_splits = 1";
    else if (E != 4096 && E != 5120) w = "n_embd is neither 4096 nor 5120 (only the LLaMA-7B/13B widths have overlapped kernel variants)";
    else if (m->hp.n_embd / m->hp.n_head != 128) w = "head dim != 128";
    if (why) *why = w;
    return w == nullptr;
}

static int build_program(thk_model* m, int seq, OvlQueue* Q) {
    thk_ctx* ctx = m->ctx;
    SeqBuf& sb = m->seqs[seq];
    if (sb.ovl_prog) return THK_OK;
    const int nl = m->l1 - m->l0;
    const int max_launches = 5 * nl + 8;
    if (!m->ovl_counters) {
        HIPCHK(ctx, hipMalloc((void**)&m->ovl_counters, (size_t)max_launches * kOvlLaunchWords * 4));
        HIPCHK(ctx, hipMemsetAsync(m->ovl_counters, 0, (size_t)max_launches * kOvlLaunchWords * 4, ctx->stream));
    }
    std::vector<OvlRecorder::Launch> store(max_launches);
    OvlRecorder rec{store.data(), 0, max_launches, false};
    m->ovl_rec = true; m->ovl_err = Q->words + 64;
    ovl_recorder = &rec;
    const int rc = enqueue_step_recorded(m, seq);
    ovl_recorder = nullptr;
    m->ovl_rec = false;
    if (rc != THK_OK) return rc;
    if (rec.overflow || rec.n < 3) return fail(ctx, THK_ERR_STATE, "overlapped dispatch: recording the step failed (%d launches)", rec.n);
    // Which packets keep the barrier bit (tunable overlap_keep_barrier, a mask over launch kinds: 1 qkv, 2 attention, 4 wo,
    // 8 w1|w3, 16 w2, 32 lm-head, 64 greedy pick; the step's first packet always does).  A launch behind a barrier packet is the
    // plain kernel; one whose own packet has no barrier bit waits inside (flavour bit 1); one whose SUCCESSOR has none writes
    // through and arrives (bit 2).  Fences follow: a barrier packet acquires at agent scope, and whatever runs before a barrier
    // packet releases at agent scope; between two launches joined by the in-kernel protocol there is no fence at all (measured:
    // with an acquire on the waiting launch the command processor starts it ~2 us later).  That is safe because after the
    // barrier packet's invalidate the only plain loads of a waiting kernel are of data nobody writes during the step (weights,
    // gains, RoPE table, embedding row, position and token); everything an earlier launch of the step wrote it reads with
    // agent-coherent loads, and every launch has made its outputs visible in memory by its end (written through, or released).
    const int keep = (int)tun(ctx, "overlap_keep_barrier");
    std::vector<int> kind(rec.n, 64), barrier(rec.n + 1, 1);
    for (int i = 0; i < rec.n; ++i) {
        int nr, u, ns, pro, epi, nsp, pipe;
        if (!strncmp(store[i].name, "thk_ovl_attn", 12)) kind[i] = 2;
        else if (sscanf(store[i].name, "thk_ovl_gemv_%d_%d_%d_%d_%d_%d_%d", &nr, &u, &ns, &pro, &epi, &nsp, &pipe) == 7)
            kind[i] = epi == GEMV_EPI_ROPE_KV ? 1 : epi == GEMV_EPI_SWIGLU ? 8 : epi == GEMV_EPI_HEAD ? 32 : pro == GEMV_PRO_ATTN ? 4 : 16;
        barrier[i] = (i == 0 || (keep & kind[i])) ? 1 : 0;
    }
    OvlProgram* P = new OvlProgram();
    std::vector<OvlKernel> ks(rec.n);
    size_t bytes = 0;
    std::vector<size_t> off(rec.n);
    for (int i = 0; i < rec.n; ++i) {
        const int flavour = (barrier[i] ? 0 : 1) | ((barrier[i + 1] || i == rec.n - 1) ? 0 : 2);
        char name[96];
        snprintf(name, sizeof name, "%s_f%d", store[i].name, flavour);
        const int r = load_kernel(ctx, Q, name, &ks[i]);
        if (r != THK_OK) { delete P; return r; }
        if ((int)ks[i].kernarg < store[i].arg_bytes) { delete P; return fail(ctx, THK_ERR_STATE, "overlapped dispatch: %s takes %u argument bytes, the host recorded %d", name, ks[i].kernarg, store[i].arg_bytes); }
        // the recorded link names both neighbours; a side that is ordered by a barrier packet is cut
        const size_t link_at = kind[i] == 2 ? offsetof(AttnArgs, ovl) : kind[i] == 64 ? offsetof(FinishArgs, ovl) : offsetof(GemvArgs, ovl);
        OvlLink* L = reinterpret_cast<OvlLink*>(store[i].args + link_at);
        if (!(flavour & 1)) { L->wait = nullptr; L->wait_n = 0; }
        if (!(flavour & 2)) L->done = nullptr;
        off[i] = bytes;
        bytes += ((size_t)ks[i].kernarg + 255) / 256 * 256;
    }
    std::vector<char> host(bytes, 0);
    for (int i = 0; i < rec.n; ++i) memcpy(host.data() + off[i], store[i].args, store[i].arg_bytes);
    if (hipMalloc(&P->kargs, bytes) != hipSuccess) { delete P; return fail(ctx, THK_ERR_OOM, "overlapped dispatch: hipMalloc(%zu) for the kernel arguments", bytes); }
    if (hipMemcpy(P->kargs, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) { hipFree(P->kargs); delete P; return fail(ctx, THK_ERR_HIP, "overlapped dispatch: uploading the kernel arguments failed"); }
    for (int i = 0; i < rec.n; ++i) {
        P->packets.push_back(make_packet(ks[i], store[i].grid, store[i].block, store[i].lds_dynamic, (char*)P->kargs + off[i]));
        bool acq = barrier[i] != 0, rel = barrier[i + 1] != 0;
        static const int fence_cut = getenv("THK_OVL_NOFENCE") ? atoi(getenv("THK_OVL_NOFENCE")) : 0;   // measurement only (results undefined): 1 no fences, 2 no releases, 3 no acquires inside a step
        if (fence_cut && i != 0) acq = acq && fence_cut == 2;
        if (fence_cut && i != rec.n - 1) rel = rel && fence_cut == 3;
        P->headers.push_back(make_header(barrier[i] != 0, acq ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE, rel ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE));
    }
    sb.ovl_prog = P;
    return THK_OK;
}

int ovl_decode_steps(thk_model* m, int seq, int n_steps) {
    thk_ctx* ctx = m->ctx;
    if (n_steps <= 0) return THK_OK;
    const char* why = nullptr;
    if (!ovl_eligible(m, &why)) return fail(ctx, THK_ERR_STATE, "overlap_dispatch is set but this model cannot use it: %s", why);
    OvlQueue* Q = nullptr;
    int rc = queue_up(ctx, &Q);
    if (rc != THK_OK) return rc;
    if (!Q->q || !Q->words) return fail(ctx, THK_ERR_STATE, "overlapped dispatch: the queue could not be set up earlier (%s)", ctx->err.c_str());
    rc = build_program(m, seq, Q);
    if (rc != THK_OK) return rc;
    const OvlProgram* P = static_cast<const OvlProgram*>(m->seqs[seq].ovl_prog);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return fail(ctx, THK_ERR_STATE, "overlapped dispatch cannot be captured into a hipGraph (its packets go to a queue of their own)");
    const unsigned k = ++Q->batch;
    hipLaunchKernelGGL(ovl_ticket_kernel, dim3(1), dim3(64), 0, ctx->stream, Q->words, k);
    HIPCHK(ctx, hipGetLastError());
    int bell = 0;
    const hsa_kernel_dispatch_packet_t gate_b = make_packet(Q->k_begin, 1, 64, 0, Q->gate_args), gate_e = make_packet(Q->k_end, 1, 64, 0, Q->gate_args);
    const uint32_t hb = make_header(true, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_NONE), he = make_header(true, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_SYSTEM);
    rc = push_packets(ctx, Q, &gate_b, &hb, 1, &bell);
    for (int s = 0; s < n_steps && rc == THK_OK; ++s) rc = push_packets(ctx, Q, P->packets.data(), P->headers.data(), P->packets.size(), &bell);
    if (rc == THK_OK) rc = push_packets(ctx, Q, &gate_e, &he, 1, &bell);
    hsa_signal_store_screlease(Q->q->doorbell_signal, (hsa_signal_value_t)(hsa_queue_load_write_index_relaxed(Q->q) - 1));
    // the stream waits for the batch: 5 s + 100 ms per step before it gives up (a 7B step takes 2.5 ms)
    hipLaunchKernelGGL(ovl_join_kernel, dim3(1), dim3(64), 0, ctx->stream, Q->words, k, (unsigned long long)(500000000ull + 10000000ull * (unsigned long long)n_steps));
    HIPCHK(ctx, hipGetLastError());
    return rc;
}

// after a stream synchronisation: did a bounded wait of the overlapped dispatch expire?
int ovl_check_error(thk_model* m) { return ovl_check_error_ctx(m->ctx); }
int ovl_check_error_ctx(thk_ctx* ctx) {
    OvlQueue* Q = static_cast<OvlQueue*>(ctx->ovl);
    if (!Q || !Q->words || !Q->batch) return THK_OK;
    unsigned e = 0;
    HIPCHK(ctx, hipMemcpyAsync(&e, Q->words + 64, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (e) {
        HIPCHK(ctx, hipMemsetAsync(Q->words + 64, 0, 4, ctx->stream));
        return fail(ctx, THK_ERR_STATE, "overlapped dispatch: a bounded wait expired (%s)",
                    e == 1 ? "a launch never saw its predecessor finish" : e == 2 ? "the batch never saw the stream's ticket" : "the stream never saw the batch finish");
    }
    return THK_OK;
}

void ovl_free_seq(SeqBuf& sb) {
    OvlProgram* P = static_cast<OvlProgram*>(sb.ovl_prog);
    if (P) { hipFree(P->kargs); delete P; }
    sb.ovl_prog = nullptr;
}

void ovl_destroy(thk_ctx* ctx) {
    OvlQueue* Q = static_cast<OvlQueue*>(ctx->ovl);
    if (!Q) return;
    if (Q->q) hsa_queue_destroy(Q->q);
    if (Q->have_exe) hsa_executable_destroy(Q->exe);
    if (Q->have_reader) hsa_code_object_reader_destroy(Q->reader);
    hipFree(Q->words); hipFree(Q->gate_args);
    if (Q->hsa_up) hsa_shut_down();
    delete Q;
    ctx->ovl = nullptr;
}

extern "C" int thk_model_uses_overlap(const thk_model* m) {
    return (m && m->finalized && m->ctx->tun.count("overlap_dispatch") && m->ctx->tun.at("overlap_dispatch") != 0 && ovl_eligible(m, nullptr)) ? 1 : 0;
}
