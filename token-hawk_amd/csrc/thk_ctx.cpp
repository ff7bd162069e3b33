// thk_ctx.cpp — context, tunables and buffers of the C-ABI in include/thk.h (thk_ctx = WGPUDevice + WGPUQueue,
// thk_buf = TensorBuffer's GPU half, th.cpp:150-229) and the helpers shared by thk_ops.cpp / thk_model.cpp.
#include "thk_internal.hpp"

// ---------------------------------------------------------------- helpers
int fail(thk_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

int64_t tun(thk_ctx* ctx, const char* name) {
    auto it = ctx->tun.find(name);
    return it == ctx->tun.end() ? 0 : it->second;
}
void default_tunables(thk_ctx* ctx) {
    ctx->tun["gemv_blocks_per_cu"] = 4;   // resident 256-thread workgroups per CU for the streaming mat-vecs
    // per-kernel launch geometry: -1 = auto (table below, from tools/sweep.py on MI355X, profiles/r01_sweep_*.json),
    // 0 = gemv_blocks_per_cu / variant 0, > 0 = explicit
    // workgroups per prefill GEMM launch (<= 256): fewer = fewer K-splits = less partial-tile traffic, but fewer CUs streaming
    for (const char* k : {"qkv", "wo", "w13", "w2"}) { ctx->tun[std::string("prefill_blocks_") + k] = 0; ctx->tun[std::string("prefill_tile_") + k] = 256; }   // blocks: 0 = auto (row-block-aligned shares where >= 192 workgroups remain, else 256); tile rows: 128 | 256
    ctx->tun["prefill_attn_mfma"] = 1;
    ctx->tun["prefill_slab_tokens"] = 256;   // prompt tokens per weight pass: 256 (eight token tiles, two half-steps per weight chunk) or 128 (rounds 1-4)
    ctx->tun["prefill_wave_grid"] = 1;       // 256-token slabs: a wave owns half the tile's rows x half the slab's tokens (gemm_prefill_v3g_kernel) instead of a quarter of the rows x all tokens (v3h)
    ctx->tun["prefill_deferred_norm"] = 1;   // RMSNorm's per-token scalar is applied on the output side of the GEMM, so the residual reducers write the next
                                             // GEMM's image themselves: 9 launches per layer instead of 11 (thk_model_prefill.cpp)
    ctx->tun["prefill_packed"] = 1;       // prefill GEMMs stream tile images of the layer matrices (a second copy of the layer weights in HBM, built on
                                          // the first prefill call) instead of 64-byte row segments of the row-major matrices
    ctx->tun["prefill_tile_wo"] = 128; ctx->tun["prefill_tile_w2"] = 128;   // 16 row-blocks only: halve the 16-way split-K partials (-3 %)
    for (const char* k : {"qkv", "wo", "w13", "w2", "head"}) {
        ctx->tun[std::string("gemv_bpc_") + k] = -1;
        ctx->tun[std::string("gemv_grid_") + k] = 0;      // > 0: this many workgroups for the launch, whatever gemv_bpc_* says (fewer than one per CU is allowed)
        ctx->tun[std::string("gemv_variant_") + k] = -1;   // (rows/iteration, slots/batch) variant, see gemv_variant()
    }
    ctx->tun["attn_splits"] = 0;          // context splits per head: 0 = auto (4; 8 when n_ctx > 1024), or 1, 2, 4, 8
    ctx->tun["attn_tc_dyn"] = 1;          // 1 = the splits partition the live context T (tc computed on the device), 0 = the cache capacity n_ctx
    ctx->tun["fold_finish"] = 1;          // the lm-head launch's last workgroup reduces the arg-max keys and finishes the token (no launch of its own)
    ctx->tun["attn_waves"] = 0;           // waves per attention block: 0 = auto (8; 16 for f32 caches longer than 1024 rows), 4 | 8 | 16
    ctx->tun["fold_embed"] = 1;           // the embedding row is fetched by layer 0's qkv prologue instead of a launch of its own
    ctx->tun["use_graph"] = 1;            // replay a captured hipGraph per decode step
    ctx->tun["measure_skip_kernel"] = 0;  // bench.py: marginal cost of one kernel = step time with minus without it (results are garbage then);
                                          // refused unless the process runs with THK_MEASURE_HOOKS=1 (never in a product)
    ctx->tun["measure_gain_alias"] = 0;   // measurement only (THK_MEASURE_HOOKS=1): the RMS prologues read the activation vector in place of the gain vector
    ctx->tun["kv_f16"] = 0;               // 1 = K/V caches stored as binary16 (half the KV bytes; k, v are rounded RNE at the append); default f32 as the reference
    ctx->tun["engine_trace"] = 0;         // development: per-op s_memtime timeline of the engine (thk_model_engine_trace)
    ctx->tun["engine"] = 0;               // 1 = decode step as ONE persistent loader/consumer launch (thk_engine.hip) when the shape allows; default 0 = 5
                                          // launches per layer: measured on MI355X the engine streams at 6.9-7.0 TB/s but every in-launch all-to-all hand-off
                                          // costs ~7 us against ~3.5 us for a kernel boundary (profiles/r02_engine_*.txt), 3.4 vs 2.5 ms per 7B token
}
// Auto launch geometry per (kernel, n_embd): {blocks per CU, variant}.  7B and 13B rows are swept values (round 3: tools/ab.py on
// the graph-replayed step, profiles/r03_ab_*.jsonl; the software-pipelined variants 5/6 won every 7B kernel but w2); other widths
// take the 7B row (their run-time slot count maps the pipelined variants back to variant 0).
Geo auto_geometry(const char* kernel, int n_embd) {
    const bool w13b = n_embd == 5120;
    if (!strcmp(kernel, "qkv")) return w13b ? Geo{1, 1} : Geo{1, 6};      // single rows, one workgroup per CU (round 4: 8 KiB chunks; RoPE pairs meet in LDS)
    if (!strcmp(kernel, "wo")) return Geo{1, 6};
    if (!strcmp(kernel, "w13")) return w13b ? Geo{8, 6} : Geo{1, 6};     // single rows of w1 | w3 alternating, one workgroup per CU (the pair meets in LDS)
    if (!strcmp(kernel, "w2")) return Geo{1, 8};                          // a workgroup per row, a wave per quarter of it (gemv_quarter_body), one workgroup per CU
    if (!strcmp(kernel, "head")) return w13b ? Geo{8, 2} : Geo{8, 6};
    return Geo{4, 0};
}
int resolve_variant(thk_ctx* ctx, const char* kernel, int n_embd) {
    const int64_t v = tun(ctx, (std::string("gemv_variant_") + kernel).c_str());
    return v < 0 ? auto_geometry(kernel, n_embd).var : (int)v;
}
int grid_for(thk_ctx* ctx, const char* specific, int n_groups, int n_embd) {
    int64_t bpc = tun(ctx, specific);
    if (bpc < 0 && !strncmp(specific, "gemv_bpc_", 9)) bpc = auto_geometry(specific + 9, n_embd).bpc;
    if (bpc <= 0) bpc = tun(ctx, "gemv_blocks_per_cu");
    if (bpc <= 0) bpc = 4;
    int64_t g = (int64_t)ctx->n_cu * bpc;
    if (!strncmp(specific, "gemv_bpc_", 9)) {
        const std::string gk = std::string("gemv_grid_") + (specific + 9);
        if (ctx->tun.count(gk) && ctx->tun[gk] > 0) g = ctx->tun[gk];
    }
    const int64_t need = (n_groups + kWaves - 1) / kWaves;
    if (g > need) g = need;
    if (g < 1) g = 1;
    return (int)g;
}
int ensure_scratch(thk_ctx* ctx, size_t bytes) {
    if (ctx->scratch_bytes >= bytes) return THK_OK;
    if (ctx->scratch) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(ctx->scratch)); ctx->scratch = nullptr; ctx->scratch_bytes = 0; }
    HIPCHK(ctx, hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    return THK_OK;
}
// RoPE table for positions [p0, p0+n): (cos, sin) of p * 10000^(-j/D), j even — the f32
// libm evaluation order of oracle orc_rope_angles (th.cpp:1476-1484).
void build_rope_table(std::vector<float>& tab, int D, int p0, int n) {
    const int half = D / 2;
    tab.resize((size_t)n * half * 2);
    for (int p = 0; p < n; ++p)
        for (int jp = 0; jp < half; ++jp) {
            const float theta = powf(10000.0f, (-(float)(2 * jp)) / (float)D);
            const float pf = (float)(p0 + p);
            tab[((size_t)p * half + jp) * 2] = cosf(pf * theta);
            tab[((size_t)p * half + jp) * 2 + 1] = sinf(pf * theta);
        }
}
static uint64_t splitmix64_h(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
uint64_t synth_key(const char* name, uint64_t seed) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (const unsigned char* p = (const unsigned char*)name; *p; ++p) { h ^= *p; h *= 0x100000001B3ull; }
    return h ^ splitmix64_h(seed);
}
float synth_scale(float sigma) { return (float)((double)sigma / 37837.2275); }

// hooks for thk_pp.cpp (same library, different translation unit)
namespace thk {
int ctx_fail(thk_ctx* ctx, int code, const char* msg) { return fail(ctx, code, "%s", msg); }
int ctx_device(thk_ctx* ctx) { return ctx ? ctx->device : 0; }
}

// ---------------------------------------------------------------- context
extern "C" int thk_abi_version(void) { return THK_ABI_VERSION; }

static int ctx_create_common(int device, hipStream_t stream, bool own, thk_ctx** out) {
    if (!out) return THK_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return THK_ERR_HIP;
    if (device < 0 || device >= count) return THK_ERR_INVALID;
    thk_ctx* ctx = new thk_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return THK_ERR_HIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        ctx->hbm_bytes = prop.totalGlobalMem;
        ctx->dev_name = prop.name;
    }
    if (own) {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return THK_ERR_HIP; }
        ctx->own_stream = true;
    } else {
        ctx->stream = stream;
    }
    default_tunables(ctx);
    *out = ctx;
    return THK_OK;
}
extern "C" int thk_ctx_create(int device_ordinal, thk_ctx** out) { return ctx_create_common(device_ordinal, nullptr, true, out); }
extern "C" int thk_ctx_create_on_stream(int device_ordinal, void* hip_stream, thk_ctx** out) {
    return ctx_create_common(device_ordinal, (hipStream_t)hip_stream, false, out);
}
extern "C" int thk_ctx_destroy(thk_ctx* ctx) {
    if (!ctx) return THK_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) hipFree(ctx->scratch);
    if (ctx->pinned_keys) hipHostFree(ctx->pinned_keys);
    if (ctx->rope_tab) hipFree(ctx->rope_tab);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return THK_OK;
}
extern "C" int thk_sync(thk_ctx* ctx) {
    if (!ctx) return THK_ERR_INVALID;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return THK_OK;
}
extern "C" const char* thk_last_error(thk_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
extern "C" void* thk_ctx_stream(thk_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
extern "C" int thk_ctx_device_info(thk_ctx* ctx, char* name, size_t name_cap, int* n_cu, size_t* hbm_bytes) {
    if (!ctx) return THK_ERR_INVALID;
    if (name && name_cap) { strncpy(name, ctx->dev_name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (n_cu) *n_cu = ctx->n_cu;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    return THK_OK;
}
extern "C" int thk_set_tunable(thk_ctx* ctx, const char* name, int64_t value) {
    if (!ctx || !name) return THK_ERR_INVALID;
    auto it = ctx->tun.find(name);
    if (it == ctx->tun.end()) return fail(ctx, THK_ERR_NOTFOUND, "unknown tunable '%s'", name);
    if ((!strcmp(name, "measure_skip_kernel") || !strcmp(name, "measure_gain_alias")) && value != 0) {
        const char* hook = getenv("THK_MEASURE_HOOKS");
        if (!hook || strcmp(hook, "1")) return fail(ctx, THK_ERR_INVALID, "%s makes a model compute garbage; it is only accepted with THK_MEASURE_HOOKS=1 in the environment", name);
    }
    it->second = value;
    return THK_OK;
}
extern "C" int thk_get_tunable(thk_ctx* ctx, const char* name, int64_t* value) {
    if (!ctx || !name || !value) return THK_ERR_INVALID;
    auto it = ctx->tun.find(name);
    if (it == ctx->tun.end()) return fail(ctx, THK_ERR_NOTFOUND, "unknown tunable '%s'", name);
    *value = it->second;
    return THK_OK;
}

// ---------------------------------------------------------------- buffers
extern "C" int thk_buf_alloc(thk_ctx* ctx, size_t bytes, thk_buf** out) {
    if (!ctx || !out) return THK_ERR_INVALID;
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    thk_buf* b = new thk_buf();
    b->size = bytes;
    hipError_t e = hipMalloc(&b->ptr, bytes ? bytes : 1);
    if (e != hipSuccess) { delete b; return fail(ctx, e == hipErrorOutOfMemory ? THK_ERR_OOM : THK_ERR_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
    e = hipMemsetAsync(b->ptr, 0, bytes ? bytes : 1, ctx->stream);
    if (e != hipSuccess) { hipFree(b->ptr); delete b; return fail(ctx, THK_ERR_HIP, "hipMemsetAsync: %s", hipGetErrorString(e)); }
    *out = b;
    return THK_OK;
}
extern "C" int thk_buf_free(thk_ctx* ctx, thk_buf* buf) {
    if (!buf) return THK_OK;
    if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); }
    if (buf->ptr) hipFree(buf->ptr);
    delete buf;
    return THK_OK;
}
extern "C" void* thk_buf_ptr(thk_buf* buf) { return buf ? buf->ptr : nullptr; }
extern "C" size_t thk_buf_size(thk_buf* buf) { return buf ? buf->size : 0; }
extern "C" int thk_buf_upload(thk_ctx* ctx, thk_buf* dst, size_t dst_off, const void* host, size_t bytes) {
    if (!ctx || !dst || (!host && bytes)) return THK_ERR_INVALID;
    REQUIRE(ctx, dst_off + bytes <= dst->size, "upload of %zu bytes at %zu exceeds buffer of %zu", bytes, dst_off, dst->size);
    HIPCHK(ctx, hipMemcpyAsync((char*)dst->ptr + dst_off, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return THK_OK;
}
extern "C" int thk_buf_download(thk_ctx* ctx, thk_buf* src, size_t src_off, void* host, size_t bytes) {
    if (!ctx || !src || (!host && bytes)) return THK_ERR_INVALID;
    REQUIRE(ctx, src_off + bytes <= src->size, "download of %zu bytes at %zu exceeds buffer of %zu", bytes, src_off, src->size);
    HIPCHK(ctx, hipMemcpyAsync(host, (const char*)src->ptr + src_off, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return THK_OK;
}
extern "C" int thk_buf_copy(thk_ctx* ctx, thk_buf* dst, size_t dst_off, thk_buf* src, size_t src_off, size_t bytes) {
    if (!ctx || !dst || !src) return THK_ERR_INVALID;
    REQUIRE(ctx, dst_off + bytes <= dst->size && src_off + bytes <= src->size, "copy range out of bounds");
    HIPCHK(ctx, hipMemcpyAsync((char*)dst->ptr + dst_off, (const char*)src->ptr + src_off, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return THK_OK;
}
