// thk_loader.cpp — GGML 'ggjt' v1 (f16) model loader of the host layer.
// Behavioural mirror of th-llama-loader.cpp (load_header :47-119, load_weights :121-265,
// post_load_init_model :330-435, load_llama_file :485-635).  File format:
//   u32 magic 0x67676a74 | u32 version 1 | i32 n_vocab,n_embd,n_mult,n_head,n_layer,n_rot,ftype
//   n_vocab x { u32 len, bytes, f32 score }
//   per tensor: i32 n_dims, i32 name_len, i32 ftype(0=f32,1=f16), i32 ne[n_dims] (ne0 = columns),
//               name, zero padding to a 32-byte FILE offset, data
// Differences from the reference: tensors go straight to the device through
// thk_model_set_tensor (no f32 host copy of tok_embeddings, no output.weight column split — the
// 256 MB WebGPU buffer limit does not exist here), n_ff is not asserted to be 11008 (Q6), and
// every malformed input returns false with a message instead of asserting.
#include "thk_host.hpp"

#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <set>

namespace th {

static const uint32_t kMagicUnversioned = 0x67676d6c;   // 'ggml'
static const uint32_t kMagicGgjt = 0x67676a74;          // 'ggjt'
static const uint32_t kFileVersion = 1;
enum { kftype_f32 = 0, kftype_f16 = 1, kftype_q40 = 2, kftype_q41 = 3 };

namespace {
struct Cursor {
    const char* p; int64_t size; int64_t off = 0; bool ok = true;
    template <typename T> T get() {
        T v{};
        if (off + (int64_t)sizeof(T) > size) { ok = false; return v; }
        memcpy(&v, p + off, sizeof(T)); off += sizeof(T);
        return v;
    }
    std::string str(int64_t n) {
        if (n < 0 || off + n > size) { ok = false; return {}; }
        std::string s(p + off, (size_t)n); off += n;
        return s;
    }
};
void fail_load(LlamaModel* m, const std::string& msg) {
    m->loadFailed = true;
    fprintf(stderr, "ERROR: %s\n", msg.c_str());
    if (m->onError) m->onError(msg);
}
}  // namespace

bool load_header(LlamaModel* m, const void* data, int64_t dataSize, int64_t* consumed) {
    Cursor c{(const char*)data, dataSize};
    const uint32_t magic = c.get<uint32_t>();
    if (!c.ok) { fail_load(m, "load_header: truncated header"); return false; }
    if (magic == kMagicUnversioned) { fail_load(m, "load_header: old unversioned ggml file (regenerate the model)"); return false; }
    if (magic != kMagicGgjt) { fail_load(m, "load_header: invalid magic value"); return false; }
    const uint32_t version = c.get<uint32_t>();
    if (!c.ok || version != kFileVersion) { fail_load(m, "load_header: unsupported file version"); return false; }
    m->n_vocab = c.get<int32_t>(); m->n_embd = c.get<int32_t>(); m->n_mult = c.get<int32_t>(); m->n_head = c.get<int32_t>();
    m->n_layer = c.get<int32_t>(); m->n_rot = c.get<int32_t>(); m->f16 = c.get<int32_t>();
    if (!c.ok || m->n_vocab <= 0 || m->n_embd <= 0 || m->n_head <= 0 || m->n_layer <= 0 || m->n_mult <= 0) {
        fail_load(m, "load_header: bad hyper-parameters"); return false;
    }
    m->vocab.id_to_token.assign((size_t)m->n_vocab, {});
    m->vocab.token_to_id.clear();
    for (int i = 0; i < m->n_vocab; ++i) {
        const uint32_t len = c.get<uint32_t>();
        if (!c.ok || len > 8096) { fail_load(m, "load_header: vocabulary entry too long or truncated"); return false; }
        std::string word = c.str(len);
        const float score = c.get<float>();
        if (!c.ok) { fail_load(m, "load_header: truncated vocabulary"); return false; }
        m->vocab.token_to_id[word] = i;
        m->vocab.id_to_token[i] = {std::move(word), score};
    }
    if (consumed) *consumed = c.off;
    return true;
}

bool parse_tensor_record(const void* data, int64_t dataSize, int64_t originalFileOffset, GgjtTensorInfo* out, std::string* err) {
    Cursor c{(const char*)data, dataSize};
    const int32_t n_dims = c.get<int32_t>(), name_len = c.get<int32_t>(), ftype = c.get<int32_t>();
    auto bad = [&](const char* why) { if (err) *err = why; return false; };
    if (!c.ok) return bad("truncated tensor header");
    if (n_dims < 1 || n_dims > 3 || name_len < 0 || name_len > 512 || ftype < 0) return bad("malformed tensor header");
    int64_t ne[3] = {1, 1, 1};
    for (int i = 0; i < n_dims; ++i) { ne[i] = c.get<int32_t>(); if (ne[i] <= 0) return bad("non-positive tensor dimension"); }
    out->name = c.str(name_len);
    if (!c.ok) return bad("truncated tensor name");
    if (ftype == kftype_f32) out->type = TensorType_F32;
    else if (ftype == kftype_f16) out->type = TensorType_F16;
    else return bad("quantized tensor types are not supported (f16 models only, README.md:5)");
    out->shape = TensorShape{};
    out->shape.c = ne[0];
    if (n_dims > 1) out->shape.r = ne[1];
    if (n_dims > 2) out->shape.b = ne[2];
    out->shape.canonicalize();
    out->ne0 = ne[0]; out->ne1 = n_dims > 1 ? ne[1] * (n_dims > 2 ? ne[2] : 1) : 1;
    const int64_t abs_off = originalFileOffset + c.off;
    const int64_t aligned = (abs_off + 31) & ~(int64_t)31;          // payload starts on a 32-byte FILE offset
    out->data_offset = c.off + (aligned - abs_off);
    out->data_bytes = ne[0] * ne[1] * ne[2] * (int64_t)get_TensorType_size(out->type);
    out->record_bytes = out->data_offset + out->data_bytes;
    if (out->record_bytes > dataSize) return bad("tensor payload exceeds the supplied buffer");
    return true;
}

static bool ensure_device_model(LlamaModel* m, thk_ctx* ctx) {
    if (m->dev) return true;
    m->ctx = ctx;
    thk_hparams hp{m->n_vocab, m->n_embd, m->n_mult, m->n_head, m->n_layer, m->n_ctx};
    if (thk_model_create(ctx, &hp, 0, m->n_layer, THK_STAGE_EMBED | THK_STAGE_HEAD, 1, &m->dev) != THK_OK) {
        fail_load(m, std::string("thk_model_create: ") + thk_last_error(ctx));
        return false;
    }
    return true;
}

bool load_weights(LlamaModel* m, thk_ctx* ctx, const void* data, int64_t dataSize, int64_t numElementsInFile, int64_t originalFileOffset) {
    if (!ensure_device_model(m, ctx)) return false;
    int64_t off = 0;
    for (int64_t i = 0; i < numElementsInFile; ++i) {
        GgjtTensorInfo ti; std::string err;
        if (!parse_tensor_record((const char*)data + off, dataSize - off, originalFileOffset + off, &ti, &err)) {
            fail_load(m, "load_weights: " + err); return false;
        }
        // as the reference (th-llama-loader.cpp:121-265): every tensor goes to the device through a TensorBuffer; the device
        // model then takes it over with one device-to-device copy and the (move-only) buffer is released at scope end
        TensorBuffer tb((const char*)data + off + ti.data_offset, ti.shape, ti.type, /*backup=*/false, ctx);
        if (!tb.is_valid() || !tb.gpu) { fail_load(m, std::string("load_weights: device upload of '") + ti.name + "' failed: " + thk_last_error(ctx)); return false; }
        const int rc = thk_model_set_tensor_dev(m->dev, ti.name.c_str(), ti.type == TensorType_F16 ? THK_F16 : THK_F32, ti.ne0, ti.ne1, tb.device_ptr());
        if (rc != THK_OK) { fail_load(m, std::string("load_weights: ") + thk_last_error(ctx)); return false; }
        m->loadedNames.push_back(ti.name);
        m->numTensorsLoaded += 1;
        off += ti.record_bytes;
    }
    if (off != dataSize) fprintf(stderr, "load_weights: %lld unknown left-over bytes\n", (long long)(dataSize - off));
    return true;
}

bool post_load_init_model(thk_ctx* ctx, std::shared_ptr<LlamaModel> m) {
    m->rng = std::mt19937(780658349);   // th-llama-loader.cpp:332-333
    if (m->loadFailed || !ensure_device_model(m.get(), ctx)) return false;
    std::set<std::string> have(m->loadedNames.begin(), m->loadedNames.end());
    std::vector<std::string> need = {"tok_embeddings.weight", "norm.weight", "output.weight"};
    static const char* per_layer[] = {"attention_norm.weight", "ffn_norm.weight", "attention.wq.weight", "attention.wk.weight",
                                      "attention.wv.weight", "attention.wo.weight", "feed_forward.w1.weight", "feed_forward.w2.weight",
                                      "feed_forward.w3.weight"};
    for (int l = 0; l < m->n_layer; ++l)
        for (const char* t : per_layer) need.push_back("layers." + std::to_string(l) + "." + t);
    for (auto& n : need)
        if (!have.count(n)) { fail_load(m.get(), "model file is missing tensor '" + n + "'"); return false; }
    if (thk_model_set_lmhead_mode(m->dev, m->lmhead_mode) != THK_OK || thk_model_finalize(m->dev) != THK_OK) {
        fail_load(m.get(), std::string("thk_model_finalize: ") + thk_last_error(ctx)); return false;
    }
    build_pipelines_llama(ctx, m);
    return true;
}

std::shared_ptr<LlamaModel> load_llama_file(thk_ctx* ctx, const std::string& filename, int lmhead_mode) {
    auto m = std::make_shared<LlamaModel>();
    m->lmhead_mode = lmhead_mode;
    FILE* f = fopen(filename.c_str(), "rb");
    if (!f) { fprintf(stderr, "Unable to open file: %s\n", filename.c_str()); return {}; }
    struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
    fseek(f, 0, SEEK_END);
    const int64_t file_size = ftell(f);
    fseek(f, 0, SEEK_SET);
    // header + vocabulary: read a growing prefix until load_header accepts it
    std::vector<char> buf;
    int64_t consumed = 0;
    {
        // magic + version are checked on the first 8 bytes BEFORE any growing read: a multi-GB file that is not ggjt v1 is
        // rejected at once instead of being pulled into RAM prefix by prefix
        char head8[8];
        if (file_size < 8 || fread(head8, 1, 8, f) != 8) { fail_load(m.get(), "load_llama_file: file too short for a ggjt header"); return {}; }
        uint32_t magic, version; memcpy(&magic, head8, 4); memcpy(&version, head8 + 4, 4);
        if (magic != kMagicGgjt || version != kFileVersion) { load_header(m.get(), head8, 8, &consumed); return {}; }   // reports "bad magic" / "bad version"
        int64_t want = std::min<int64_t>(file_size, 1 << 20);
        for (;;) {
            buf.resize((size_t)want);
            fseek(f, 0, SEEK_SET);
            if ((int64_t)fread(buf.data(), 1, (size_t)want, f) != want) return {};
            LlamaModel probe;   // quiet probe: only its success matters
            probe.onError = [](std::string) {};
            if (load_header(&probe, buf.data(), want, &consumed)) break;
            if (want == file_size) { load_header(m.get(), buf.data(), want, &consumed); return {}; }   // report the real error
            want = std::min<int64_t>(file_size, want * 4);
        }
        if (!load_header(m.get(), buf.data(), (int64_t)buf.size(), &consumed)) return {};
    }
    // tensors: read each record [header | padding | data] and hand it to load_weights, as the reference does (:571-621)
    int64_t pos = consumed;
    while (pos < file_size) {
        char hdr[12 + 3 * 4 + 512] = {0};
        fseek(f, pos, SEEK_SET);
        const size_t got = fread(hdr, 1, sizeof hdr, f);
        if (got < 12) break;
        int32_t n_dims, name_len;
        memcpy(&n_dims, hdr, 4); memcpy(&name_len, hdr + 4, 4);
        if (n_dims < 1 || n_dims > 3 || name_len < 0 || name_len > 512) { fail_load(m.get(), "load_llama_file: malformed tensor header"); return {}; }
        const int64_t head = 12 + 4 * n_dims + name_len;
        if ((int64_t)got < head) { fail_load(m.get(), "load_llama_file: truncated tensor header"); return {}; }
        int32_t ftype; memcpy(&ftype, hdr + 8, 4);
        const int64_t elt = ftype == kftype_f32 ? 4 : ftype == kftype_f16 ? 2 : 0;
        if (elt == 0) { fail_load(m.get(), "load_llama_file: quantized formats are not supported"); return {}; }
        // every dimension positive, and the element count bounded by what the file can hold BEFORE multiplying: a negative
        // or huge dim can neither slip past the size check as a negative record length nor overflow the int64 product
        int64_t count = 1;
        bool dims_ok = true;
        for (int i = 0; i < n_dims; ++i) {
            int32_t v; memcpy(&v, hdr + 12 + 4 * i, 4);
            if (v <= 0 || count > file_size / elt / v) { dims_ok = false; break; }
            count *= v;
        }
        if (!dims_ok) { fail_load(m.get(), "load_llama_file: tensor dimensions are non-positive or exceed the file size"); return {}; }
        const int64_t data_at = (pos + head + 31) & ~(int64_t)31;
        const int64_t rec = (data_at - pos) + count * elt;
        if (rec <= 0 || pos + rec > file_size) { fail_load(m.get(), "load_llama_file: truncated tensor data"); return {}; }
        buf.resize((size_t)rec);
        fseek(f, pos, SEEK_SET);
        if ((int64_t)fread(buf.data(), 1, (size_t)rec, f) != rec) { fail_load(m.get(), "load_llama_file: short read"); return {}; }
        if (!load_weights(m.get(), ctx, buf.data(), rec, 1, pos)) return {};
        pos += rec;
    }
    if (!post_load_init_model(ctx, m)) return {};
    return m;
}

}  // namespace th
