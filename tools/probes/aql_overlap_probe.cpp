// aql_overlap_probe.cpp — round-3 probe: what does a dependent chain of weight-streaming kernels gain if the AQL packets carry
// NO barrier bit and the dependency is enforced inside the kernels (arrival counter), so that the next kernel's waves are
// resident with their first weight batch in flight while the previous kernel's last waves finish?  (HIP always sets the bit
// for work of one stream; this is the hardware analogue of programmatic dependent launch.)
//
// The chain imitates a LLaMA-7B decode layer at 4096 columns: 100.7 MB (qkv), 33.6 MB (wo), 180.4 MB (w13), 90.2 MB (w2-sized),
// each kernel reading a vector its predecessor wrote, over `layers` distinct weight sets (> the 256 MB memory-side cache).
//   mode A  barrier bit on every packet, plain loads (what a hipGraph replay does)
//   mode B  no barrier bit, in-kernel wait on the predecessor's arrival counter, agent-scope vector loads / stores
//   mode C  barrier bit AND the mode-B kernel (cost of the in-kernel protocol alone)
// Kernels are loaded from aql_probe_kernels.hsaco with the HSA runtime and dispatched on a user-mode queue.
// Build: g++ -O2 -std=c++17 aql_overlap_probe.cpp -I/opt/rocm/include -L/opt/rocm/lib -lhsa-runtime64 -o aql_overlap_probe
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <string>
#include <vector>

#define HSACHK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS && s_ != HSA_STATUS_INFO_BREAK) { const char* m_ = nullptr; hsa_status_string(s_, &m_); printf("%s failed: %s\n", #x, m_ ? m_ : "?"); exit(1); } } while (0)

struct ProbeArgs {            // must match aql_probe_kernels.hip
    const uint16_t* W; int n_groups, n_blocks; const float* x_in; float* y_out; const unsigned* prev_done; unsigned prev_target; unsigned* my_done; int wait_mode; unsigned* err; int n_shards, poll_sleep, first_sleeps;
};

static hsa_agent_t g_gpu; static bool g_have_gpu = false;
static hsa_amd_memory_pool_t g_dev_pool, g_karg_pool; static bool g_have_dev = false, g_have_karg = false;
static hsa_status_t find_gpu(hsa_agent_t a, void*) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t find_dev_pool(hsa_amd_memory_pool_t p, void*) {
    hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t flags; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    bool alloc; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (!alloc) return HSA_STATUS_SUCCESS;
    if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_have_dev) { g_dev_pool = p; g_have_dev = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t find_karg_pool_cpu(hsa_agent_t a, void*) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t != HSA_DEVICE_TYPE_CPU || g_have_karg) return HSA_STATUS_SUCCESS;
    hsa_amd_agent_iterate_memory_pools(a, [](hsa_amd_memory_pool_t p, void*) -> hsa_status_t {
        hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
        if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
        uint32_t flags; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
        if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_have_karg) { g_karg_pool = p; g_have_karg = true; }
        return HSA_STATUS_SUCCESS; }, nullptr);
    return HSA_STATUS_SUCCESS;
}
static void* dev_alloc(size_t bytes) { void* p = nullptr; HSACHK(hsa_amd_memory_pool_allocate(g_dev_pool, bytes, 0, &p)); return p; }

struct Kernel { uint64_t object; uint32_t kernarg_size, group_size, private_size; };
static Kernel get_kernel(hsa_executable_t exe, const char* name) {
    hsa_executable_symbol_t sym; HSACHK(hsa_executable_get_symbol_by_name(exe, name, &g_gpu, &sym));
    Kernel k{};
    HSACHK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    HSACHK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg_size));
    HSACHK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group_size));
    HSACHK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.private_size));
    return k;
}

static hsa_queue_t* g_q;
static void dispatch(const Kernel& k, void* kernarg, uint32_t grid_threads, bool barrier, int acq, int rel, hsa_signal_t done) {
    const uint64_t idx = hsa_queue_add_write_index_relaxed(g_q, 1);
    while (idx - hsa_queue_load_read_index_scacquire(g_q) >= g_q->size) { }
    hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)g_q->base_address + (idx & (g_q->size - 1));
    p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->reserved0 = 0;
    p->grid_size_x = grid_threads; p->grid_size_y = 1; p->grid_size_z = 1;
    p->private_segment_size = k.private_size; p->group_segment_size = k.group_size;
    p->kernel_object = k.object; p->kernarg_address = kernarg; p->reserved2 = 0; p->completion_signal = done;
    const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                            (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
    __atomic_store_n((uint32_t*)p, (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
    hsa_signal_store_screlease(g_q->doorbell_signal, (hsa_signal_value_t)idx);
}

int main(int argc, char** argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 8, reps = argc > 2 ? atoi(argv[2]) : 10;
    const char* hsaco = argc > 3 ? argv[3] : "aql_probe_kernels.hsaco";
    HSACHK(hsa_init());
    HSACHK(hsa_iterate_agents(find_gpu, nullptr));
    if (!g_have_gpu) { printf("no GPU agent\n"); return 1; }
    HSACHK(hsa_amd_agent_iterate_memory_pools(g_gpu, find_dev_pool, nullptr));
    HSACHK(hsa_iterate_agents(find_karg_pool_cpu, nullptr));
    if (!g_have_dev || !g_have_karg) { printf("memory pools not found\n"); return 1; }
    // code object
    FILE* f = fopen(hsaco, "rb"); if (!f) { printf("cannot open %s\n", hsaco); return 1; }
    fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> blob(sz); if (fread(blob.data(), 1, sz, f) != (size_t)sz) return 1; fclose(f);
    hsa_code_object_reader_t reader; HSACHK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &reader));
    hsa_profile_t prof; HSACHK(hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_PROFILE, &prof));
    hsa_executable_t exe; HSACHK(hsa_executable_create_alt(prof, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HSACHK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    HSACHK(hsa_executable_freeze(exe, nullptr));
    const Kernel k = get_kernel(exe, "gemv_dep.kd");
    printf("gemv_dep: kernarg %u B, LDS %u B, scratch %u B\n", k.kernarg_size, k.group_size, k.private_size);
    HSACHK(hsa_queue_create(g_gpu, 16384, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &g_q));
    // data
    const int C = 4096;
    const int groups[4] = {6144, 2048, 11008, 5504};               // row pairs: 100.7, 33.6, 180.4, 90.2 MB
    const int blocks[4] = {768, 256, 1024, 512};
    const int nk = layers * 4;
    std::vector<uint16_t*> W(nk);
    for (int i = 0; i < nk; ++i) { const size_t bytes = (size_t)groups[i % 4] * 2 * C * 2; W[i] = (uint16_t*)dev_alloc(bytes); HSACHK(hsa_amd_memory_fill(W[i], 0x3c003c00u, bytes / 4)); }
    float* vec[2]; for (int i = 0; i < 2; ++i) { vec[i] = (float*)dev_alloc(1 << 20); HSACHK(hsa_amd_memory_fill(vec[i], 0x3a000000u, (1 << 20) / 4)); }
    const int kShardStride = 64 * 32;                               // words per kernel: up to 64 shards, 128 bytes apart
    unsigned* counters = (unsigned*)dev_alloc((size_t)(nk + 1) * kShardStride * 4); HSACHK(hsa_amd_memory_fill(counters, 0, (size_t)(nk + 1) * kShardStride));
    unsigned* err = counters + (size_t)nk * kShardStride;
    const int n_shards = getenv("PROBE_SHARDS") ? atoi(getenv("PROBE_SHARDS")) : 16, poll_sleep = getenv("PROBE_SLEEP") ? atoi(getenv("PROBE_SLEEP")) : 1;
    const int first_sleeps = getenv("PROBE_FIRST") ? atoi(getenv("PROBE_FIRST")) : 0, acq_b = getenv("PROBE_ACQ") ? atoi(getenv("PROBE_ACQ")) : 1;
    printf("shards %d, poll sleep %d, first sleeps %d, mode-B acquire scope %d\n", n_shards, poll_sleep, first_sleeps, acq_b);
    const size_t kslot = (k.kernarg_size + 63) / 64 * 64;
    char* kargs = nullptr; HSACHK(hsa_amd_memory_pool_allocate(g_karg_pool, kslot * nk * (size_t)(reps + 2) * 6, 0, (void**)&kargs));
    HSACHK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, kargs));
    hsa_signal_t done; HSACHK(hsa_signal_create(1, 0, nullptr, &done));
    // kernel arguments live in DEVICE memory (as HIP places them): built on the host per run, copied over before the clock starts
    char* kargs_dev = (char*)dev_alloc(kslot * nk * (size_t)(reps + 2));
    const bool dev_kargs = getenv("PROBE_HOST_KERNARG") == nullptr;
    unsigned epoch = 0;                                              // completed repetitions of the counter-based modes
    auto run = [&](int mode, int n_rep) -> double {
        std::vector<char> host(kslot * nk * (size_t)n_rep, 0);
        struct P { void* karg; uint32_t grid; bool barrier, last; };
        std::vector<P> pk;
        for (int r = 0; r < n_rep; ++r) {
            for (int i = 0; i < nk; ++i) {
                ProbeArgs a{};
                a.W = W[i]; a.n_groups = groups[i % 4]; a.n_blocks = blocks[i % 4]; a.x_in = vec[i & 1]; a.y_out = vec[(i + 1) & 1];
                a.wait_mode = mode == 0 ? 0 : 1; a.err = err;
                if (a.wait_mode) {
                    const int prev = i == 0 ? nk - 1 : i - 1;
                    // counters only grow: kernel i has been run (epoch + r) times before, its predecessor one more time (or as often, for i == 0)
                    a.prev_done = (r == 0 && i == 0) ? nullptr : counters + (size_t)prev * kShardStride;
                    a.prev_target = (unsigned)((i == 0 ? epoch + r : epoch + r + 1) * (unsigned)blocks[prev % 4]);
                    a.my_done = counters + (size_t)i * kShardStride;
                }
                a.n_shards = n_shards; a.poll_sleep = poll_sleep; a.first_sleeps = first_sleeps;
                const size_t off = kslot * ((size_t)r * nk + i);
                memcpy(host.data() + off, &a, sizeof a);
                pk.push_back(P{(dev_kargs ? kargs_dev : kargs) + off, (uint32_t)a.n_blocks * 256, mode != 1 || (r == 0 && i == 0), r == n_rep - 1 && i == nk - 1});
            }
        }
        if (dev_kargs) HSACHK(hsa_memory_copy(kargs_dev, host.data(), host.size())); else memcpy(kargs, host.data(), host.size());
        hsa_signal_store_relaxed(done, 1);
        const auto t0 = std::chrono::steady_clock::now();
        for (const P& p : pk)
            dispatch(k, p.karg, p.grid, p.barrier, mode == 1 ? acq_b : HSA_FENCE_SCOPE_AGENT, mode == 1 ? HSA_FENCE_SCOPE_NONE : HSA_FENCE_SCOPE_AGENT, p.last ? done : hsa_signal_t{0});
        while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) { }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (mode != 0) epoch += (unsigned)n_rep;
        return us / n_rep;
    };
    const double bytes_per_rep = 0; (void)bytes_per_rep;
    double mb = 0; for (int i = 0; i < 4; ++i) mb += (double)groups[i] * 2 * C * 2 / 1e6;
    const char* names[3] = {"A barrier bit, plain kernel", "B NO barrier bit, in-kernel arrival counters", "C barrier bit + in-kernel counters"};
    for (int mode : {0, 2, 1, 0, 1}) {
        run(mode, 2);
        const double us = run(mode, reps);
        unsigned e = 0; HSACHK(hsa_memory_copy(&e, err, 4));
        printf("mode %s: %.1f us per %d-layer pass = %.2f us per layer (%.0f MB/layer -> %.2f TB/s)%s\n", names[mode], us, layers, us / layers, mb, mb * layers / us,
               e ? "  [TIMEOUT in a wait]" : "");
        if (e) { HSACHK(hsa_amd_memory_fill(err, 0, 1)); }
    }
    return 0;
}
