#!/bin/bash
# Development: shader clock and socket power while the prefill chain / the decode loop runs (rocm-smi sampled from the side).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== idle"; rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power" | head -8
python - <<'PY' &
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as graft
thk = graft.load_package()
with thk.Context(0) as ctx:
    m = thk.Model(ctx, thk.LLAMA_7B); m.fill_synthetic(); m.finalize()
    toks = np.concatenate([[1], np.random.default_rng(1).integers(3, 32000, 127)]).astype(np.int32)
    m.prefill(toks, 0)
    print("PHASE prefill t=%.2f" % (time.time() % 100000), flush=True)
    t0 = time.time()
    while time.time() - t0 < 8: m.prefill(toks, 0, want_logits=False)
    print("PHASE decode t=%.2f" % (time.time() % 100000), flush=True)
    m.seq_set(0, 1, 511)
    t0 = time.time()
    while time.time() - t0 < 8: m.decode_steps(64, 0, advance=False)
    print("PHASE done t=%.2f" % (time.time() % 100000), flush=True)
PY
for i in $(seq 1 26); do echo "== t=$(date +%s.%N | cut -c7-14) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket" | sed 's/.*: //' | tr '\n' ' ')"; sleep 0.7; done
wait
