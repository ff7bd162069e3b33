"""N>1 path on CPU: the pipeline driver (token-hawk_amd/pipeline.py) with world_size 2 and 3
over gloo.  The stage compute is supplied by the oracle (the HIP stage needs a GPU); what is
under test is the schedule, the P2P pattern and the token feedback ring."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleStage:
    def __init__(self, orc, shape, rank, world, n_seq, layer_range):
        self.orc, self.shape = orc, shape
        self.l0, self.l1 = layer_range(shape.n_layer, rank, world)
        self.is_first, self.is_last = rank == 0, rank == world - 1
        self.m = orc.OracleModel(shape, n_seq)
        self.m.fill_synthetic()
        E = shape.n_embd
        self.hidden_in = [torch.zeros(E) for _ in range(n_seq)]
        self.hidden_out = [torch.zeros(E) for _ in range(n_seq)]
        self.token = [torch.zeros(1, dtype=torch.int32) for _ in range(n_seq)]
        self.pos = [0] * n_seq
        self.gen = [[] for _ in range(n_seq)]
        self.logits = [[] for _ in range(n_seq)]

    def set_seq(self, s, token, pos):
        self.token[s][0] = token; self.pos[s] = pos

    def set_token(self, s, token):
        self.token[s][0] = int(token)

    def step(self, s, advance):
        if self.is_first:
            lg, hid = self.m.eval(int(self.token[s][0]), self.pos[s], seq=s, l0=self.l0, l1=self.l1, want_logits=self.is_last)
        else:
            lg, hid = self.m.eval(None, self.pos[s], seq=s, l0=self.l0, l1=self.l1, hidden=self.hidden_in[s].numpy(),
                                  want_logits=self.is_last)
        if self.is_last:
            t = self.orc.greedy(lg); self.token[s][0] = t; self.gen[s].append(t); self.logits[s].append(lg.copy())
        else:
            self.hidden_out[s].copy_(torch.from_numpy(hid))
        if advance:
            self.pos[s] += 1


def _worker(rank, world, port, n_prompt, n_gen, steady=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as graft
        from oracle import oracle as orc
        thk = graft.load_package()
        from token_hawk_amd.pipeline import PipelineDriver, layer_range
        shape = orc.ModelShape(n_vocab=2048, n_embd=512, n_mult=256, n_head=8, n_layer=3 if world == 3 else 4, n_ctx=64)
        S = world
        stage = OracleStage(orc, shape, rank, world, S, layer_range)
        drv = PipelineDriver(stage, rank, world, S)
        rng = np.random.default_rng(7)
        prompts = rng.integers(3, 2048, (n_prompt, S)); prompts[0, :] = 1
        for s in range(S):
            stage.set_seq(s, int(prompts[0, s]), 0)
        r1 = drv.run(n_prompt, advance=True, forced_tokens=prompts)       # prompt through the pipeline
        assert r1.items == n_prompt * S
        extra = [0] * S
        if not steady:
            r2 = drv.run(n_gen, advance=True)                              # greedy continuation (token ring)
            assert r2.items == n_gen * S
        else:
            # the ring kept full across calls (what bench.py times): prime, two steady() calls, drain.  Every rank processes exactly
            # steps * S items per steady() call; after the drain all N - 1 + n_gen * S issued items have left the last stage, so the
            # first N - 1 sequences are one token ahead of the others.
            p = drv.prime(advance=True)
            a = drv.steady(1, advance=True)
            b = drv.steady(n_gen - 1, advance=True)
            d = drv.drain(advance=True)
            assert a.items == S and a.micro_steps == S and b.items == (n_gen - 1) * S and b.micro_steps == (n_gen - 1) * S
            assert p.items + d.items == world - 1 and p.micro_steps == d.micro_steps == world - 1
            extra = [1 if s < world - 1 else 0 for s in range(S)]
            r3 = drv.run(1, advance=True)                                  # an empty ring takes a self-contained run again
            assert r3.items == S
            extra = [e + 1 for e in extra]
        if rank == world - 1:
            full = orc.OracleModel(shape, S); full.fill_synthetic()
            for s in range(S):
                for i in range(n_prompt):
                    lg, _ = full.eval(int(prompts[i, s]), i, seq=s)
                exp, tok = [], orc.greedy(lg)
                exp.append(tok)
                assert (stage.logits[s][n_prompt - 1] == lg).all()
                for i in range(n_gen + extra[s]):
                    lg, _ = full.eval(tok, n_prompt + i, seq=s); tok = orc.greedy(lg); exp.append(tok)
                    assert (stage.logits[s][n_prompt + i] == lg).all(), (s, i)
                got = stage.gen[s]
                # gen holds one token per processed item: n_prompt (incl. the one after the last prompt token) + n_gen
                assert got[n_prompt - 1:] == exp, (s, got[n_prompt - 1:], exp)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipeline_ring_matches_single_process(world):
    port = 29500 + world + (os.getpid() % 1000)
    mp.spawn(_worker, args=(world, port, 3, 4), nprocs=world, join=True)


@pytest.mark.parametrize("world", [2, 3])
def test_pipeline_ring_kept_full_matches_single_process(world):
    """prime / steady / steady / drain (no fill or drain inside a steady() call) produces the same tokens and logits."""
    port = 29600 + world + (os.getpid() % 1000)
    mp.spawn(_worker, args=(world, port, 3, 4, True), nprocs=world, join=True)


def test_layer_range_partition():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    graft.load_package()
    from token_hawk_amd.pipeline import layer_range
    for L in (32, 40, 3):
        for N in (1, 2, 3, 4, 8):
            if N > L:
                continue
            r = [layer_range(L, k, N) for k in range(N)]
            assert r[0][0] == 0 and r[-1][1] == L and all(r[i][1] == r[i + 1][0] for i in range(N - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    assert layer_range(32, 3, 8) == (12, 16)
