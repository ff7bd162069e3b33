#!/bin/bash
# dev: prefill GEMM on packed tile images vs row-major weights: same logits bit for bit, time
mkdir -p gpurun_out/r02
cd /root/repo
for pk in 0 1; do
  timeout 300 python tools/bench_prefill.py 7b 128 x prefill_packed=$pk > gpurun_out/r02/pk$pk.json 2> gpurun_out/r02/pk$pk.err
  tail -1 gpurun_out/r02/pk$pk.json
done
timeout 300 python tools/bench_prefill.py 7b 512 x prefill_packed=1 2>&1 | tail -1
timeout 300 python tools/bench_prefill.py 13b 128 x prefill_packed=1 2>&1 | tail -1
timeout 400 python - <<'PY' > gpurun_out/r02/pk_parity.txt 2>&1
import numpy as np, __graft_entry__ as graft
thk = graft.load_package()
shape = thk.LLAMA_7B
out = {}
for M in (128, 96, 33, 7, 300):
    rng = np.random.default_rng(M)
    toks = np.concatenate([[1], rng.integers(3, shape.n_vocab, M - 1)]).astype(np.int32)
    for pk in (0, 1):
        with thk.Context(0) as ctx:
            ctx.set_tunable("prefill_packed", pk)
            m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
            out[(M, pk)] = m.prefill(toks, 0).copy()
            m.close()
    d = np.abs(out[(M, 0)] - out[(M, 1)]).max()
    print("M", M, "max |row-major - packed| logits", d, "finite", np.isfinite(out[(M, 1)]).all())
PY
cat gpurun_out/r02/pk_parity.txt
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -q -m gpu -k "prefill or gemm" 2>&1 | tail -5
