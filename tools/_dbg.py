import sys, time, numpy as np
sys.path.insert(0,'.')
import __graft_entry__ as g
thk = g.load_package()
T=512
with thk.Context(0) as ctx:
    m = thk.Model(ctx, thk.LLAMA_7B); m.fill_synthetic()
    def run(tag, **tun):
        for k,v in tun.items(): ctx.set_tunable(k,v)
        m.finalize()
        m.eval(np.arange(3,11,dtype=np.int32), 0, want_logits=False); m.seq_set(0,5,T-1)
        agg={}
        for _ in range(4):
            for k,ms in m.profile_step(0): agg.setdefault(k,[]).append(ms*1e3)
        best=0
        for rep in range(3):
            for _ in range(5): m.decode_step(0, False)
            ctx.sync(); t0=time.perf_counter()
            for _ in range(60): m.decode_step(0, False)
            ctx.sync(); best=max(best, 60/(time.perf_counter()-t0))
        print(tag, tun, "tok/s %.1f"%best, {k: round(float(np.mean(v)),1) for k,v in agg.items() if 'attn' in k or 'wo' in k}, flush=True)
    run("base", attn_combine=0, attn_splits=4)
    for sp in (2,4,8):
        for bpc,var in ((2,0),(1,3),(4,0),(2,1)):
            run("combine", attn_combine=1, attn_splits=sp, gemv_bpc_wo=bpc, gemv_variant_wo=var)
