import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
thk = g.load_package()
with thk.Context(0) as ctx:
    m = thk.Model(ctx, thk.LLAMA_7B); m.fill_synthetic(); m.finalize()
    m.seq_set(0, 5, 511)
    m.prepare_steps(5); m.decode_steps(5, 0, advance=False); ctx.sync()
    for n in (20, 24, 28):
        m.prepare_steps(n); ctx.sync()
        ts = []
        for rep in range(4):
            t0 = time.perf_counter(); m.decode_steps(n, 0, advance=False); ctx.sync(); ts.append((time.perf_counter() - t0) / n * 1e3)
        print(n, " ".join(f"{t:.4f}" for t in ts), flush=True)
    m.close()
