"""Quick engine bring-up check: tiny model logits vs oracle with the engine on/off, then a 7B timing."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft
from oracle import oracle as orc
orc.build()
thk = graft.load_package()
ctx = thk.Context(0)
what = sys.argv[1:] or ["tiny", "7b"]
if "tiny" in what:
    for shape_name in ("TINY", "TINY_Q1"):
        for eng in (0, 1):
            ctx.set_tunable("engine", eng)
            m = thk.Model(ctx, getattr(thk, shape_name)); m.fill_synthetic(); m.finalize()
            om = orc.OracleModel(getattr(orc, shape_name)); om.fill_synthetic()
            rng = np.random.default_rng(1)
            toks = [1] + rng.integers(3, 2048, 20).tolist()
            worst = 0.0; worst_h = 0.0
            try:
                for i, t in enumerate(toks):
                    lg, hid = m.eval([t], i, want_hidden=True)
                    lo, ho = om.eval(t, i, flags=orc.FAITHFUL_ORDER)
                    worst = max(worst, float(np.abs(lg - lo).max())); worst_h = max(worst_h, float(np.abs(hid - ho).max()))
                    if not np.isfinite(lg).all() or np.abs(lg - lo).max() > 1e-3:
                        print(f"  MISMATCH at token {i}: max|dlogit|={np.abs(lg - lo).max():.3e} argmax {int(lg.argmax())} vs {orc.greedy(lo)}")
                        break
                m.seq_set(0, 7, len(toks))
                m.decode_steps(9, 0, advance=True)
                gen, n, pos = m.seq_get(0)
                tok, exp = 7, []
                for i in range(9):
                    lo, _ = om.eval(tok, len(toks) + i); tok = orc.greedy(lo); exp.append(tok)
                print(f"{shape_name} engine={eng} uses_engine={m.uses_engine()} max|dlogit|={worst:.3e} max|dhidden|={worst_h:.3e} greedy_ok={gen.tolist() == exp} pos={pos}")
            except Exception as e:
                print(f"{shape_name} engine={eng}: EXCEPTION {e}")
            m.close(); om.close()
if "7b" in what or "13b" in what:
    shape = thk.LLAMA_13B if "13b" in what else thk.LLAMA_7B
    for eng in (0, 1):
        ctx.set_tunable("engine", eng)
        m = thk.Model(ctx, shape); m.fill_synthetic(); m.finalize()
        T = 512
        m.seq_set(0, 5, 0)
        m.decode_steps(T - 1, 0, advance=True)      # fill the cache with the greedy continuation
        ctx.sync()
        m.prepare_steps(100)
        m.decode_steps(10, 0, advance=False); ctx.sync()
        t0 = time.perf_counter(); m.decode_steps(100, 0, advance=False); ctx.sync(); dt = time.perf_counter() - t0
        gen, n, pos = m.seq_get(0)
        print(f"{shape.name if hasattr(shape,'name') else 'model'} engine={eng} uses_engine={m.uses_engine()} {100 / dt:.1f} tok/s {dt * 10:.4f} ms/step pos={pos} tail={gen[-4:].tolist()}")
        if eng == 0: ref_tail = gen[-8:].tolist()
        else: print("   greedy tail equal to launch path:", gen[-8:].tolist() == ref_tail)
        try:
            prof = m.profile_step(0)
            print("   eager profile:", ", ".join(f"{k}={v*1e3:.1f}us" for k, v in prof if k in ("engine", "embed", "finish_token")) if eng else "")
        except Exception as e:
            print("   profile failed:", e)
        m.close()
ctx.close()
