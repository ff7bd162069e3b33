import sys, numpy as np
sys.path.insert(0,'.')
import __graft_entry__ as g
from oracle import oracle as orc
thk = g.load_package()
with thk.Context(0) as ctx:
    n = 1<<18
    d = ctx.alloc(n*2)
    ctx.synth_f16("tok_embeddings.weight", n, d)
    a = d.download(np.uint16, n); b = orc.synth_f16("tok_embeddings.weight", orc.TENSOR_SEED, orc.TENSOR_SIGMA, n)
    bad = np.nonzero(a != b)[0]
    print("mismatch", len(bad), "of", n)
    print(bad[:10], a[bad[:10]], b[bad[:10]], a[bad[:10]].view(np.float16), b[bad[:10]].view(np.float16))
    gd = ctx.alloc(4096*4); ctx.synth_gain_f32("norm.weight", 4096, gd)
    ga = gd.download(np.float32, 4096); gb = orc.synth_gain("norm.weight", orc.TENSOR_SEED, orc.TENSOR_SIGMA, 4096)
    print("gain mismatch", (ga != gb).sum())
