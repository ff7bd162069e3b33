#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -k "context_beyond" 2>&1 | grep -v "^$" | tail -30
timeout 300 python - <<'PY'
import numpy as np, __graft_entry__ as graft
thk = graft.load_package()
shape = thk.ModelShape(n_vocab=2048, n_embd=512, n_mult=256, n_head=8, n_layer=2, n_ctx=1200)
rng = np.random.default_rng(1100)
toks = np.concatenate([[1], rng.integers(3, 2048, 1102)]).astype(np.int32)
out = {}
for n in (76, 128, 204, 1100):
    for pk in (0, 1):
        with thk.Context(0) as ctx:
            ctx.set_tunable("prefill_packed", pk)
            m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
            lp = m.prefill(toks[:n], 0).copy()
            m.reset_kv(0)
            ld, _ = m.eval(toks[:n], 0)
            print("n", n, "packed", pk, "max |prefill - token by token|", float(np.abs(lp - ld).max()), "finite", bool(np.isfinite(lp).all()))
            m.close()
PY
