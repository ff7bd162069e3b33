#!/bin/bash
# dev: prefill GEMM v4 (loader waves) against v3: same logits bit for bit, time
mkdir -p gpurun_out/r02
cd /root/repo
for lw in 0 1; do
  timeout 300 python tools/bench_prefill.py 7b 128 x prefill_loader_waves=$lw > gpurun_out/r02/v4_lw$lw.json 2> gpurun_out/r02/v4_lw$lw.err
  tail -1 gpurun_out/r02/v4_lw$lw.json
done
timeout 300 python - <<'PY' > gpurun_out/r02/v4_parity.txt 2>&1
import numpy as np, __graft_entry__ as graft
thk = graft.load_package()
shape = thk.LLAMA_7B
out = {}
for M in (128, 96, 33, 7):
    rng = np.random.default_rng(M)
    toks = np.concatenate([[1], rng.integers(3, shape.n_vocab, M - 1)]).astype(np.int32)
    for lw in (0, 1):
        with thk.Context(0) as ctx:
            ctx.set_tunable("prefill_loader_waves", lw)
            m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
            out[(M, lw)] = m.prefill(toks, 0).copy()
            m.close()
    d = np.abs(out[(M, 0)] - out[(M, 1)]).max()
    print("M", M, "max |v3 - v4| logits", d, "finite", np.isfinite(out[(M, 1)]).all())
PY
cat gpurun_out/r02/v4_parity.txt
