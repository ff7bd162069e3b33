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
        self.bulk = torch.zeros(shape.n_ctx, E)
        self.pos = [0] * n_seq
        self.gen = [[] for _ in range(n_seq)]
        self.logits = [[] for _ in range(n_seq)]

    def set_seq(self, s, token, pos):
        self.token[s][0] = token; self.pos[s] = pos

    def prefill(self, s, tokens, n, n_past):
        """HipStage.prefill's contract on the oracle: rows [0, n) of self.bulk in place, the last stage's pick into the token slot, position -> n_past + n
        (the oracle has no batched pass: token by token through this stage's layers - the result the MFMA pass must reproduce)."""
        for i in range(n):
            last_row = i == n - 1
            lg, hid = self.m.eval(int(tokens[i]) if self.is_first else None, n_past + i, seq=s, l0=self.l0, l1=self.l1,
                                  hidden=None if self.is_first else self.bulk[i].numpy(), want_logits=self.is_last and last_row)
            if not self.is_last:
                self.bulk[i].copy_(torch.from_numpy(hid))
            elif last_row:
                t = self.orc.greedy(lg); self.token[s][0] = t; self.gen[s].append(t); self.logits[s].append(lg.copy())
            else:
                self.gen[s].append(None); self.logits[s].append(None)      # keeps one entry per prompt position, as the ring fill does
        self.pos[s] = n_past + n

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


def _worker(rank, world, port, n_prompt, n_gen, steady=False, split=None, validate=False, prefill=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as graft
        from oracle import oracle as orc
        thk = graft.load_package()
        from token_hawk_amd.pipeline import PipelineDriver, layer_range
        n_layer = sum(split) if split else (3 if world == 3 else 4)
        shape = orc.ModelShape(n_vocab=2048, n_embd=512, n_mult=256, n_head=8, n_layer=n_layer, n_ctx=64)
        S = world
        ranges = (lambda L, r, N: (sum(split[:r]), sum(split[:r + 1]))) if split else layer_range      # an uneven split (balanced_layer_split)
        stage = OracleStage(orc, shape, rank, world, S, ranges)
        drv = PipelineDriver(stage, rank, world, S)
        if validate:      # what bench.py does before anything is timed: a known pattern through every slot of the chosen transport
            rep = drv.validate_handoff(reps=3)
            assert rep.ok and rep.checked == 2 * S and rep.handoff_us > 0, rep
        rng = np.random.default_rng(7)
        prompts = rng.integers(3, 2048, (n_prompt, S)); prompts[0, :] = 1
        for s in range(S):
            stage.set_seq(s, int(prompts[0, s]), 0)
        if prefill:      # one batched pass per stage and sequence, rows handed forward as bulk messages, the pick fed back to rank 0
            r1 = drv.prefill(prompts)
            assert r1.items == S and stage.pos == [n_prompt] * S
        else:
            r1 = drv.run(n_prompt, advance=True, forced_tokens=prompts)       # prompt through the pipeline
            assert r1.items == n_prompt * S
        extra = [0] * S
        if not steady:
            r2 = drv.run(n_gen, advance=True)                              # greedy continuation (token ring)
            assert r2.items == n_gen * S
        else:
            # the ring kept full across calls (what bench.py times): prime, two steady() calls, drain.  Every rank processes exactly
            # steps * S items per steady() call; prime() issues N - 1 items beyond whole steps and drain() tops the ring up to the
            # step boundary (one more item when S == N), so every sequence ends exactly one token ahead and forced tokens work again.
            p = drv.prime(advance=True)
            a = drv.steady(1, advance=True)
            b = drv.steady(n_gen - 1, advance=True)
            d = drv.drain(advance=True)
            assert a.items == S and a.micro_steps == S and b.items == (n_gen - 1) * S and b.micro_steps == (n_gen - 1) * S
            assert d.topped_up == 1 and p.items + d.items == S and p.micro_steps == world - 1 and d.micro_steps == world
            assert drv.base % S == 0
            r3 = drv.run(1, advance=True, forced_tokens=None)              # an empty ring takes a self-contained run again
            assert r3.items == S
            extra = [2] * S
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


@pytest.mark.parametrize("world,steady", [(2, False), (3, False), (3, True)])
def test_pipeline_prompt_pass_per_stage_matches_single_process(world, steady):
    """PipelineDriver.prefill (round 6: one batched prompt pass per stage, M x E rows handed forward, pick fed back) followed by the token
    ring produces the tokens and logits of the un-split model - the same check the ring-fed prompt passes."""
    port = 29650 + 2 * world + int(steady) + (os.getpid() % 1000)
    mp.spawn(_worker, args=(world, port, 5, 4, steady, None, False, True), nprocs=world, join=True)


@pytest.mark.parametrize("world,split", [(2, (3, 1)), (3, (1, 2, 1))])
def test_uneven_layer_split_matches_single_process(world, split):
    """Stages of different depth (what balanced_layer_split hands to HipStage): tokens and logits of the un-split model, after
    the hand-off pattern check bench.py runs first."""
    port = 29700 + world + (os.getpid() % 1000)
    mp.spawn(_worker, args=(world, port, 3, 4, True, split, True), nprocs=world, join=True)


def _validate_worker(rank, world, port, corrupt_rank):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("thk_pipeline_only", os.path.join(ROOT, "token-hawk_amd", "pipeline.py"))
        pipe = importlib.util.module_from_spec(spec); sys.modules["thk_pipeline_only"] = pipe; spec.loader.exec_module(pipe)

        class Slots:                                      # a stage is only its buffers here
            def __init__(self):
                self.is_first, self.is_last = rank == 0, rank == world - 1
                self.hidden_in = [torch.zeros(64) for _ in range(world)]
                self.hidden_out = [torch.zeros(64) for _ in range(world)]
                self.token = [torch.zeros(1, dtype=torch.int32) for _ in range(world)]
        st = Slots()
        drv = pipe.PipelineDriver(st, rank, world, world)
        if rank == corrupt_rank:                          # a transport that delivers one wrong word / a wrong token on this rank
            orig = drv._xfer

            def bad(send_seq, recv_seq):
                orig(send_seq, recv_seq)
                if recv_seq == 1:
                    if st.is_first:
                        st.token[1] += 1
                    else:
                        st.hidden_in[1][5] += 1.0
            drv._xfer = bad
        rep = drv.validate_handoff(reps=2)
        ok = torch.tensor([1 if rep.ok else 0]); dist.all_reduce(ok, op=dist.ReduceOp.MIN)      # how bench.py makes the ranks agree
        if corrupt_rank < 0:
            assert rep.ok and not rep.errors and int(ok) == 1
        else:
            assert int(ok) == 0 and (rep.ok != (rank == corrupt_rank))
            if rank == corrupt_rank:
                assert len(rep.errors) == 2 and "seq 1" in rep.errors[0]
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("corrupt_rank", [-1, 0, 1])
def test_handoff_validation_catches_a_bad_transport(corrupt_rank):
    """validate_handoff: known patterns through every (sequence, kind) slot; a receiver that gets one wrong hidden-state word
    (rank 1) or a wrong token (rank 0) reports it and the all-reduced verdict fails on every rank."""
    port = 29800 + corrupt_rank + (os.getpid() % 1000)
    mp.spawn(_validate_worker, args=(3, port, corrupt_rank), nprocs=3, join=True)


def _dying_worker(rank, world, port, q):
    import datetime
    import time
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=15))
    import importlib.util
    spec = importlib.util.spec_from_file_location("thk_pipeline_only", os.path.join(ROOT, "token-hawk_amd", "pipeline.py"))
    pipe = importlib.util.module_from_spec(spec); sys.modules["thk_pipeline_only"] = pipe; spec.loader.exec_module(pipe)

    class Stage:
        def __init__(self):
            self.is_first, self.is_last = rank == 0, rank == world - 1
            self.hidden_in = [torch.zeros(64) for _ in range(world)]
            self.hidden_out = [torch.zeros(64) for _ in range(world)]
            self.token = [torch.zeros(1, dtype=torch.int32) for _ in range(world)]
            self.n = 0

        def step(self, s, advance):
            self.n += 1
            if rank == 1 and self.n == 5:
                os._exit(0)                               # this rank dies in the middle of the steady ring
    drv = pipe.PipelineDriver(Stage(), rank, world, world)
    t0 = time.time()
    try:
        with pipe.Watchdog(40.0, "steady ring with a dying peer"):
            drv.prime(advance=False)
            drv.steady(200, advance=False)
        q.put(("completed", time.time() - t0))
    except Exception as e:                                # gloo reports the closed connection / the group's time-out
        q.put(("error", time.time() - t0, type(e).__name__))


def test_a_dead_rank_is_an_error_not_a_hang():
    """Rank 1 dies inside steady(): rank 0's bounded waits surface as an exception (gloo: connection closed or the group's 15 s
    time-out) or, failing that, the watchdog ends the process with exit code 3 - never a hang."""
    import time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_dying_worker, args=(r, 2, port, q)) for r in range(2)]
    t0 = time.time()
    for p in procs:
        p.start()
    procs[0].join(timeout=90)
    alive = procs[0].is_alive()
    for p in procs:
        if p.is_alive():
            p.kill()
        p.join()
    assert not alive, "rank 0 was still waiting for a dead peer after 90 s"
    assert time.time() - t0 < 90
    res = q.get(timeout=5) if procs[0].exitcode == 0 else None
    assert (res is not None and res[0] == "error") or procs[0].exitcode == 3, (res, procs[0].exitcode)


def test_balanced_layer_split():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    graft.load_package()
    from token_hawk_amd.pipeline import balanced_layer_split, layer_range, split_efficiency_bound, stage_cost
    t_layer, t_head = 75.7, 45.0                          # LLaMA-7B on MI355X, us: the lm-head costs 0.6 of a layer
    for N in (1, 2, 4, 8):
        sp = balanced_layer_split(32, N, t_layer, t_head)
        assert sp == [layer_range(32, r, N) for r in range(N)]          # uniform is already optimal: BASELINE's 32/16/8/4
        assert sp[0][0] == 0 and sp[-1][1] == 32 and all(sp[i][1] == sp[i + 1][0] for i in range(N - 1))
    assert abs(split_efficiency_bound(balanced_layer_split(32, 8, t_layer, t_head), t_layer, t_head) - 0.887) < 2e-3
    # a head as heavy as two layers moves layers off the last rank, and the bound rises
    heavy = balanced_layer_split(32, 8, 1.0, 2.1)
    uni = [layer_range(32, r, 8) for r in range(8)]
    assert heavy[-1][1] - heavy[-1][0] < 4 and sum(b - a for a, b in heavy) == 32 and min(b - a for a, b in heavy) >= 1
    assert split_efficiency_bound(heavy, 1.0, 2.1) > split_efficiency_bound(uni, 1.0, 2.1) + 0.05
    # exhaustive check of optimality on a small case
    import itertools
    best = min(max(stage_cost(n, r == 0, r == 2, 1.0, 1.7, 0.3) for r, n in enumerate(c))
               for c in itertools.product(range(1, 8), repeat=3) if sum(c) == 9)
    got = balanced_layer_split(9, 3, 1.0, 1.7, 0.3)
    assert abs(max(stage_cost(b - a, r == 0, r == 2, 1.0, 1.7, 0.3) for r, (a, b) in enumerate(got)) - best) < 1e-9


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
