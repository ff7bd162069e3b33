// thk_cli.cpp — native front-end with the reference CLI's contract (cli/main.cpp:26-198):
//   th -m <ggml-model-f16.bin> "<prompt>"        (-h for help; -d <dir>, the dead chunked
//   format of README.md:61, is rejected).  Device bring-up is thk_ctx_create instead of the
//   Dawn instance/adapter/device dance (cli/main.cpp:72-105).
#include <stdio.h>
#include <string.h>
#include <string>

#include "thk_host.hpp"

static void usage(const char* argv0) {
    fprintf(stderr, "usage: %s [-m model.bin] [--greedy] [--prefill] [--faithful-lmhead] [--device N] \"prompt\"\n", argv0);
}

int main(int argc, char** argv) {
    th::SamplerParams sp;
    std::string model = "models/7B/ggml-model-f16.bin", prompt;
    int device = 0, lm_mode = THK_LMHEAD_CORRECT;
    bool prefill = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-h" || a == "--help") { usage(argv[0]); return 0; }
        else if (a == "-m" && i + 1 < argc) model = argv[++i];
        else if (a == "-d") { fprintf(stderr, "the chunked model directory format was removed upstream; pass a ggjt file with -m\n"); return 2; }
        else if (a == "--greedy") sp.temp = 0.0f;
        else if (a == "--prefill") prefill = true;
        else if (a == "--faithful-lmhead") lm_mode = THK_LMHEAD_FAITHFUL;
        else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (!a.empty() && a[0] == '-') { fprintf(stderr, "unknown option %s\n", a.c_str()); usage(argv[0]); return 2; }
        else prompt = a;
    }
    if (prompt.empty()) { usage(argv[0]); return 2; }
    thk_ctx* ctx = nullptr;
    if (thk_ctx_create(device, &ctx) != THK_OK) { fprintf(stderr, "no usable HIP device %d (libthk has no CPU fallback)\n", device); return 1; }
    auto m = th::load_llama_file(ctx, model, lm_mode);
    if (!m) { fprintf(stderr, "failed to load %s\n", model.c_str()); thk_ctx_destroy(ctx); return 1; }
    m->sampler = sp;
    m->prefillPrompt = prefill;
    int n = 0;
    const double t0 = th::get_time_seconds();
    m->onNewToken = [&](std::string tok, std::string) { fputs(tok.c_str(), stdout); fflush(stdout); ++n; };
    m->onError = [](std::string e) { fprintf(stderr, "\nerror: %s\n", e.c_str()); };
    th::do_inference(ctx, m, prompt);
    const double dt = th::get_time_seconds() - t0;
    fprintf(stderr, "\n[%d tokens generated, %d positions, %.1f positions/s]\n", n, m->n_past, m->n_past / dt);
    m.reset();
    thk_ctx_destroy(ctx);
    return 0;
}
