"""GPU-side checks of the multi-GPU plumbing that can run on ONE GPU: zero-copy torch views of
libthk device memory, libthk on a torch stream, and the pipeline driver's RCCL point-to-point calls
with a single-rank ring (rank 0 is both first and last stage and sends the token to itself).
The N>1 schedule itself is covered on CPU by tests/test_pipeline_gloo.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_torch_view_is_zero_copy(thk):
    import torch
    from token_hawk_amd.pipeline import _CudaView
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = thk.Context(0, stream=stream.cuda_stream)
        buf = ctx.alloc(4096 * 4)
        t = torch.as_tensor(_CudaView(buf.ptr, (4096,), "<f4"), device=torch.device("cuda", 0))
        assert t.data_ptr() == buf.ptr and t.dtype == torch.float32
        t.copy_(torch.arange(4096, dtype=torch.float32, device="cuda"))
        stream.synchronize()
        assert (buf.download(np.float32, 4096) == np.arange(4096, dtype=np.float32)).all()
        buf.upload(np.full(4096, 7.0, np.float32))
        assert float(t.sum().item()) == 7.0 * 4096
        ti = torch.as_tensor(_CudaView(buf.ptr, (1,), "<i4"), device=torch.device("cuda", 0))
        assert ti.dtype == torch.int32
        buf.free(); ctx.close()


def test_single_rank_ring_over_rccl(thk, orc):
    """HipStage + PipelineDriver with backend nccl (= RCCL) and world_size 1: every micro-step posts a
    grouped self send/recv of the token through batch_isend_irecv on libthk-owned memory."""
    import torch
    import torch.distributed as dist
    from token_hawk_amd.pipeline import HipStage, PipelineDriver
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            ctx = thk.Context(0, stream=stream.cuda_stream)
            stage = HipStage(thk, ctx, thk.TINY, 0, 1, 1, dev)
            drv = PipelineDriver(stage, 0, 1, 1, force_ring=True)
            prompt = np.array([[1], [40], [900]], np.int32)
            stage.set_seq(0, 1, 0)
            drv.run(3, advance=True, forced_tokens=prompt)
            drv.run(5, advance=True)
            torch.cuda.synchronize(dev)
            got = stage.generated(0)
            om = orc.OracleModel(orc.TINY); om.fill_synthetic()
            for i, t in enumerate(prompt[:, 0].tolist()):
                lg, _ = om.eval(t, i)
            tok, exp = orc.greedy(lg), []
            exp.append(tok)
            for i in range(5):
                lg, _ = om.eval(tok, 3 + i); tok = orc.greedy(lg); exp.append(tok)
            assert got[2:] == exp
            stage.model.close(); ctx.close()
    finally:
        dist.destroy_process_group()


def test_native_rccl_hand_off_between_two_stages(thk, orc, ctx):
    """thk_pp_* (the C-ABI's own RCCL point-to-point path, no torch): two layer-range stages on one
    GPU, the hidden state travels stage A -> stage B through a grouped ncclSend/ncclRecv to self,
    the greedy token travels back the same way; logits must equal the un-split model's."""
    import ctypes as C
    lib = ctx.lib
    uid = C.create_string_buffer(128)
    assert lib.thk_pp_get_unique_id(uid) == 0
    pp = C.c_void_p()
    ctx.check(lib.thk_pp_create(ctx.h, 1, 0, uid, C.byref(pp)), "thk_pp_create")
    assert lib.thk_pp_size(pp) == 1 and lib.thk_pp_rank(pp) == 0
    # raw buffers
    a, b = ctx.from_numpy(np.arange(4096, dtype=np.float32)), ctx.alloc(4096 * 4)
    ctx.check(lib.thk_pp_group_begin(pp), "group_begin")
    ctx.check(lib.thk_pp_send(pp, C.c_void_p(a.ptr), 4096 * 4, 0), "send")
    ctx.check(lib.thk_pp_recv(pp, C.c_void_p(b.ptr), 4096 * 4, 0), "recv")
    ctx.check(lib.thk_pp_group_end(pp), "group_end")
    ctx.sync()
    assert (b.download(np.float32, 4096) == np.arange(4096, dtype=np.float32)).all()
    # two stages + full model
    shape = thk.TINY
    full = thk.Model(ctx, shape); full.fill_synthetic(); full.finalize()
    sa = thk.Model(ctx, shape, 0, 1, flags=thk.THK_STAGE_EMBED); sa.fill_synthetic(); sa.finalize()
    sb = thk.Model(ctx, shape, 1, 2, flags=thk.THK_STAGE_HEAD); sb.fill_synthetic(); sb.finalize()
    full.seq_set(0, 1, 0); sa.seq_set(0, 1, 0); sb.seq_set(0, 1, 0)
    for step in range(6):
        full.decode_step(0, True)
        sa.decode_step(0, True)
        ctx.check(lib.thk_pp_group_begin(pp), "group_begin")
        ctx.check(lib.thk_pp_send_hidden(pp, sa.h, 0, 0), "send_hidden")
        ctx.check(lib.thk_pp_recv_hidden(pp, sb.h, 0, 0), "recv_hidden")
        ctx.check(lib.thk_pp_group_end(pp), "group_end")
        sb.decode_step(0, True)
        ctx.check(lib.thk_pp_group_begin(pp), "group_begin")       # token ring: last stage -> first stage
        ctx.check(lib.thk_pp_send_token(pp, sb.h, 0, 0), "send_token")
        ctx.check(lib.thk_pp_recv_token(pp, sa.h, 0, 0), "recv_token")
        ctx.check(lib.thk_pp_group_end(pp), "group_end")
    gf, nf, pf = full.seq_get(0)
    gb, nb, pb = sb.seq_get(0)
    assert nf == nb == 6 and pf == pb == 6 and gf.tolist() == gb.tolist()
    # round 6: the prompt pass per stage - the M x E rows stage A leaves travel to stage B's buffer as ONE thk_pp message (a bandwidth message: 0.8 MB here,
    # 8 MB in the second round below), stage B finishes the pass; logits of the un-split prompt pass
    E, M = shape.n_embd, 40
    toks = np.concatenate([[1], np.random.default_rng(5).integers(3, shape.n_vocab, M - 1)]).astype(np.int32)
    rows_a, rows_b = thk.Buffer(ctx, M * E * 4), thk.Buffer(ctx, M * E * 4)
    for mm in (full, sa, sb):
        mm.reset_kv(0)
    lf = full.prefill(toks, 0)
    sa.prefill_stage(toks, rows_a, M, 0)
    ctx.check(lib.thk_pp_group_begin(pp), "group_begin")
    ctx.check(lib.thk_pp_send(pp, C.c_void_p(rows_a.ptr), M * E * 4, 0), "send rows")
    ctx.check(lib.thk_pp_recv(pp, C.c_void_p(rows_b.ptr), M * E * 4, 0), "recv rows")
    ctx.check(lib.thk_pp_group_end(pp), "group_end")
    ls = sb.prefill_stage(None, rows_b, M, 0, want_logits=True)
    assert np.abs(ls - lf).max() < 1e-4 and int(ls.argmax()) == int(lf.argmax())
    big = np.random.default_rng(6).standard_normal(2 * 1024 * 1024).astype(np.float32)      # 8 MB = 511 x 4096 x 4: the 7B prompt's message size
    src, dst = ctx.from_numpy(big), ctx.alloc(big.nbytes)
    ctx.check(lib.thk_pp_group_begin(pp), "group_begin")
    ctx.check(lib.thk_pp_send(pp, C.c_void_p(src.ptr), big.nbytes, 0), "send 8 MB")
    ctx.check(lib.thk_pp_recv(pp, C.c_void_p(dst.ptr), big.nbytes, 0), "recv 8 MB")
    ctx.check(lib.thk_pp_group_end(pp), "group_end")
    ctx.sync()
    assert (dst.download(np.float32, big.size) == big).all()
    assert lib.thk_pp_destroy(pp) == 0
    for m in (full, sa, sb):
        m.close()


def test_driver_with_native_transport_single_rank_ring(thk, orc):
    """PipelineDriver on the native thk_pp_* transport (world 1, forced ring): same tokens as the oracle."""
    import ctypes as C
    import torch
    from token_hawk_amd.pipeline import HipStage, PipelineDriver
    dev = torch.device("cuda", 0)
    ctx = thk.Context(0)
    stage = HipStage(thk, ctx, thk.TINY, 0, 1, 1, dev)
    uid = C.create_string_buffer(128)
    assert ctx.lib.thk_pp_get_unique_id(uid) == 0
    stage.attach_native_transport(0, 1, uid.raw)
    drv = PipelineDriver(stage, 0, 1, 1, force_ring=True)
    prompt = np.array([[1], [40], [900]], np.int32)
    stage.set_seq(0, 1, 0)
    drv.run(3, advance=True, forced_tokens=prompt)
    drv.run(5, advance=True)
    got = stage.generated(0)
    om = orc.OracleModel(orc.TINY); om.fill_synthetic()
    for i, t in enumerate(prompt[:, 0].tolist()):
        lg, _ = om.eval(t, i)
    tok, exp = orc.greedy(lg), []
    exp.append(tok)
    for i in range(5):
        lg, _ = om.eval(tok, 3 + i); tok = orc.greedy(lg); exp.append(tok)
    assert got[2:] == exp
    ctx.lib.thk_pp_destroy(stage.pp)
    stage.model.close(); ctx.close()


@pytest.mark.parametrize("transport", ["auto", "native", "torch", "peer"])
def test_bench_pipeline_path_end_to_end_on_one_gpu(transport):
    """`bench.py --force-pipeline --model tiny`: the N>1 code path of the benchmark itself (process group on RCCL, HipStage, ring
    kept full across prime / steady / drain, stage timing, JSON line) runs end to end with one rank and a self send/recv, and the
    line carries the keys the N>1 runs report."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env["MASTER_PORT"] = str(29800 + os.getpid() % 150 + {"native": 0, "torch": 1, "peer": 2, "auto": 3}[transport])
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-pipeline", "--model", "tiny", "--steps", "6", "--warmup", "2",
                        "--transport", transport, "--no-cpu-baseline", "--no-kernel-profile"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["ranks_joined"] == 1 and d["steps"] == 6 and d["value"] > 0
    if transport == "auto":                                           # the chain native -> peer -> torch: the first that validates, with the reasons
        assert d["config"]["transport"] in ("native", "peer", "torch") and d["handoff"]["transport_log"]
    else:
        assert d["config"]["transport"] == transport                 # a named transport is tried alone (the run fails if it does not validate)
    h = d["handoff"]                                                  # the pattern round trip every boundary passed before anything was timed
    assert h["validated"] is True and h["payloads_checked_per_rank"] == 2 and h["handoff_us"] > 0 and len(h["handoff_us_per_rank"]) == 1
    ls = d["layer_split"]
    assert ls["used"] == ls["uniform"] == ls["balanced"] == [[0, 2]] and 0 < ls["bound_uniform"] <= 1.0
    for key in ("single_stream", "stage_ms_no_handoff", "ideal_pipeline_tokens_per_s", "pure_replica_upper_bound_tokens_per_s",
                "ideal_efficiency_bound", "timed_region"):
        assert key in d, key
    assert len(d["stage_ms_no_handoff"]) == 1 and 0 < d["ideal_efficiency_bound"] <= 1.0
    assert "extras" not in d and "cpu_baseline" not in d              # N>1 protocol: no single-GPU extras on the pipeline path


def test_driver_with_peer_mailbox_transport_single_rank_ring(thk, orc):
    """PipelineDriver on the mailbox transport (thk_peer_*, world 1: the stage is its own successor, no IPC): the token is
    pushed into the stage's own mailbox and waited for by the next micro-step; same tokens as the oracle, no time-out."""
    import torch
    from token_hawk_amd.pipeline import HipStage, PipelineDriver
    dev = torch.device("cuda", 0)
    ctx = thk.Context(0)
    stage = HipStage(thk, ctx, thk.TINY, 0, 1, 1, dev)
    handle = stage.attach_peer_transport(1)
    assert len(handle) == 64
    stage.connect_peer(None)
    drv = PipelineDriver(stage, 0, 1, 1, force_ring=True)
    prompt = np.array([[1], [40], [900]], np.int32)
    stage.set_seq(0, 1, 0)
    drv.run(3, advance=True, forced_tokens=prompt)
    drv.prime(advance=True); drv.steady(5, advance=True); drv.drain(advance=True)
    stage.peer_check()
    got = stage.generated(0)
    om = orc.OracleModel(orc.TINY); om.fill_synthetic()
    for i, t in enumerate(prompt[:, 0].tolist()):
        lg, _ = om.eval(t, i)
    tok, exp = orc.greedy(lg), []
    exp.append(tok)
    for i in range(5):
        lg, _ = om.eval(tok, 3 + i); tok = orc.greedy(lg); exp.append(tok)
    assert got[2:] == exp
    ctx.lib.thk_peer_destroy(stage.peer)
    stage.model.close(); ctx.close()


def test_peer_bulk_slot_round_trip_and_argument_errors(thk):
    """thk_peer_send_bulk / thk_peer_recv_bulk (round 6) on a single-stage ring (the stage is its own successor): the rows of a prompt pass go through the sequence's
    bulk slot of the mailbox and come back bit for bit, twice (flags only grow), for two sequences; sizes that are not multiples of 16 bytes, exceed the slot
    (n_ctx x n_embd x 4) or come before thk_peer_connect are refused with a message."""
    import ctypes as C
    import torch
    from token_hawk_amd.pipeline import HipStage
    dev = torch.device("cuda", 0)
    ctx = thk.Context(0)
    lib = ctx.lib
    stage = HipStage(thk, ctx, thk.TINY, 0, 1, 2, dev)
    stage.attach_peer_transport(2)
    E, n_ctx = thk.TINY.n_embd, thk.TINY.n_ctx
    src, dst = thk.Buffer(ctx, n_ctx * E * 4), thk.Buffer(ctx, n_ctx * E * 4)
    assert lib.thk_peer_send_bulk(stage.peer, 0, C.c_void_p(src.ptr), 1024) != 0 and b"connect" in lib.thk_last_error(ctx.h)
    stage.connect_peer(None)
    rng = np.random.default_rng(3)
    for rnd in range(2):
        for seq, rows in ((0, n_ctx), (1, 17)):
            a = rng.standard_normal(rows * E).astype(np.float32)
            src.upload(a)
            dst.upload(np.zeros(rows * E, np.float32))
            ctx.check(lib.thk_peer_send_bulk(stage.peer, seq, C.c_void_p(src.ptr), a.nbytes), "send_bulk")
            ctx.check(lib.thk_peer_recv_bulk(stage.peer, seq, C.c_void_p(dst.ptr), a.nbytes), "recv_bulk")
            stage.peer_check()
            assert np.array_equal(dst.download(np.float32, rows * E), a), (rnd, seq)
    for bad in (24, n_ctx * E * 4 + 16, 0):
        assert lib.thk_peer_send_bulk(stage.peer, 0, C.c_void_p(src.ptr), bad) != 0
        assert lib.thk_peer_recv_bulk(stage.peer, 0, C.c_void_p(dst.ptr), bad) != 0
    assert lib.thk_peer_send_bulk(stage.peer, 2, C.c_void_p(src.ptr), 1024) != 0      # no such sequence
    stage.peer_check()                                                                 # refused calls enqueue nothing and raise no flag
    lib.thk_peer_destroy(stage.peer)
    stage.model.close(); ctx.close()


def test_bench_two_ranks_share_one_gpu():
    """`bench.py --gpus 2 --ranks-share-gpu`: the benchmark's whole N = 2 flow with REAL stages in two processes (self-launch through
    torch.distributed.run, two HipStages of the tiny model on cuda:0, process group over gloo, transport chosen by the hand-off
    pattern check = IPC mailboxes, prime / steady / drain, MAX over ranks, stage times gathered, one JSON line) - everything the
    first multi-GPU run does except the hop across xGMI."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--ranks-share-gpu", "--model", "tiny", "--steps", "6", "--warmup", "2",
                        "--no-cpu-baseline", "--no-kernel-profile", "--master-port", str(29650 + os.getpid() % 200)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_joined"] == 2 and d["steps"] == 6 and d["value"] > 0 and "plumbing_check" in d
    assert d["n1_same_invocation"]["n_gpus"] == 1 and d["n1_same_invocation"]["value"] > 0
    assert d["config"]["transport"] == "peer" and d["config"]["sequences"] == 2
    h = d["handoff"]
    assert h["validated"] is True and h["payloads_checked_per_rank"] == 4 and len(h["handoff_us_per_rank"]) == 2
    ls = d["layer_split"]
    assert ls["used"] == [[0, 1], [1, 2]] and "measured" in ls and len(d["stage_ms_no_handoff"]) == 2
    kf = d["kv_fill"]                                                 # round 6: the prompts reach the caches both ways, and the caches agree
    assert kf["ring_s"] > 0 and kf["prefill_s"] > 0 and kf["first_greedy_token_equal"] is True and kf["ring_revolutions"] == 63 and kf["prefill_passes_per_stage"] == 2


def test_bench_four_ranks_share_one_gpu_and_carry_the_n1_line():
    """`bench.py --gpus 4 --ranks-share-gpu --model tiny4`: four processes, four one-layer HipStages on cuda:0 (the deepest ring the GPU
    suite can build), and the N = 1 line of the same invocation on rank 0 (`n1_same_invocation`: what SCALE's first point must agree with)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--ranks-share-gpu", "--model", "tiny4", "--steps", "6", "--warmup", "2",
                        "--no-cpu-baseline", "--no-kernel-profile", "--master-port", str(29850 + os.getpid() % 100)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["ranks_joined"] == 4 and d["value"] > 0 and d["config"]["sequences"] == 4 and d["config"]["transport"] == "peer"
    assert d["layer_split"]["used"] == [[0, 1], [1, 2], [2, 3], [3, 4]] and len(d["stage_ms_no_handoff"]) == 4
    assert d["handoff"]["validated"] is True
    assert d["kv_fill"]["first_greedy_token_equal"] is True and d["kv_fill"]["prefill_s"] > 0
    n1 = d["n1_same_invocation"]
    assert n1["n_gpus"] == 1 and n1["steps"] == 6 and n1["value"] > 0


def test_peer_timeout_is_sticky(thk):
    """A hand-off that never arrives: the bounded wait (~2 s) raises the error word instead of hanging the GPU, thk_peer_check
    reports it - and keeps reporting it - every later send / recv is refused (the flag sequence is out of step from then on), a
    wait already enqueued behind the failure returns at once; a fresh peer works again.  The mailbox's memory kind is reported."""
    import time
    import torch
    from token_hawk_amd.pipeline import HipStage
    dev = torch.device("cuda", 0)
    ctx = thk.Context(0)
    stage = HipStage(thk, ctx, thk.TINY, 0, 1, 1, dev)
    stage.attach_peer_transport(1)
    stage.connect_peer(None)
    assert ctx.lib.thk_peer_memory_kind(stage.peer) in (0, 1, 2)
    lib, peer = ctx.lib, stage.peer
    t0 = time.time()
    assert lib.thk_peer_recv(peer, 0, 0) == 0          # nobody sent: this wait times out on the device
    assert lib.thk_peer_recv(peer, 0, 1) == 0          # enqueued behind it: must not wait another 2 s
    assert lib.thk_peer_check(peer) != 0               # THK_ERR_STATE
    assert 1.0 < time.time() - t0 < 3.8, time.time() - t0
    assert b"timed out" in lib.thk_last_error(ctx.h)
    assert lib.thk_peer_check(peer) != 0               # sticky
    assert lib.thk_peer_send(peer, 0, 0) != 0 and lib.thk_peer_recv(peer, 0, 0) != 0
    assert b"destroy" in lib.thk_last_error(ctx.h)
    lib.thk_peer_destroy(peer)
    stage.attach_peer_transport(1); stage.connect_peer(None)      # a new peer on the same stage: counters and flags start afresh
    stage.peer_exchange([("hidden", 0)], [("hidden", 0)])
    stage.peer_check()
    lib.thk_peer_destroy(stage.peer)
    stage.model.close(); ctx.close()


def _peer_worker(rank, world, port, n_prompt, n_gen, q, prefill=False):
    """One pipeline stage per PROCESS, both on GPU 0: the hidden state and the token cross the process boundary through
    hipIpc-mapped mailboxes (thk_peer_*); gloo only carries the 64-byte handles and the barriers."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as graft
        thk = graft.load_package()
        from token_hawk_amd.pipeline import HipStage, PipelineDriver
        dev = torch.device("cuda", 0)
        shape = thk.ModelShape(n_vocab=2048, n_embd=512, n_mult=256, n_head=8, n_layer=4, n_ctx=max(64, n_prompt + n_gen + 8))
        S = world
        ctx = thk.Context(0)
        stage = HipStage(thk, ctx, shape, rank, world, S, dev)
        handles = [None] * world
        dist.all_gather_object(handles, stage.attach_peer_transport(S))
        stage.connect_peer(handles[(rank + 1) % world])
        dist.barrier()
        drv = PipelineDriver(stage, rank, world, S)
        # what bench.py does before anything is timed: known patterns through every mailbox slot, both ways round the ring
        rep = drv.validate_handoff(reps=4, sync=ctx.sync)
        stage.peer_check()
        assert rep.ok and rep.checked == 2 * S and rep.handoff_us > 0, rep
        assert ctx.lib.thk_peer_memory_kind(stage.peer) in (0, 1, 2)
        print(f"[peer r{rank}] mailbox memory kind {ctx.lib.thk_peer_memory_kind(stage.peer)} (0 coarse, 1 uncached, 2 fine-grained), bare hand-off {rep.handoff_us:.1f} us", flush=True)
        rng = np.random.default_rng(7)
        prompts = rng.integers(3, 2048, (n_prompt, S)); prompts[0, :] = 1
        for s in range(S):
            stage.set_seq(s, int(prompts[0, s]), 0)
        if prefill:      # round 6: ONE MFMA prompt pass per stage and sequence, the M x E rows through the mailbox's bulk slot, the pick fed back
            drv.prefill(prompts)
        else:
            drv.run(n_prompt, advance=True, forced_tokens=prompts)
        drv.prime(advance=True); drv.steady(n_gen, advance=True); drv.drain(advance=True)
        ctx.sync()
        stage.peer_check()
        dist.barrier()
        if rank == world - 1:
            full = thk.Model(ctx, shape, n_seq=S); full.fill_synthetic(); full.finalize()
            ok = True
            for s in range(S):
                full.seq_set(s, int(prompts[0, s]), 0)
                if prefill:
                    lp = full.prefill(prompts[:, s].astype(np.int32), 0, seq=s)
                    exp = [int(lp.argmax())]
                else:
                    full.eval(prompts[:, s].astype(np.int32), 0, seq=s, want_logits=False)      # logs the greedy pick after every prompt token
                    exp = full.seq_get(s)[0].tolist()
                extra = 1                                                                 # prime issues N - 1 items beyond whole steps, drain() tops the step up: every sequence ends one token ahead
                full.seq_set(s, exp[-1], n_prompt)
                full.decode_steps(n_gen + extra, s, advance=True)
                exp += full.seq_get(s)[0].tolist()                                        # n_prompt + n_gen + extra picks
                got = stage.generated(s)
                if prefill:
                    exp = exp[1:]                                                         # the pick behind the prompt lives in the token slot only; the log starts with the first ring step
                ok = ok and got == exp
                if got != exp:
                    print("MISMATCH", s, got, exp, flush=True)
            full.close()
            q.put(bool(ok))
        dist.barrier()
        ctx.lib.thk_peer_destroy(stage.peer)
        stage.model.close(); ctx.close()
    finally:
        dist.destroy_process_group()


def test_two_process_pipeline_through_peer_mailboxes_on_one_gpu():
    """A REAL two-stage HipStage pipeline: two processes (one per stage, both on GPU 0), layers 0-1 + embedding in one, layers
    2-3 + lm-head in the other, two sequences in flight, ring kept full (prime / steady / drain).  Hidden states and tokens cross
    the process boundary through IPC-mapped device mailboxes (no RCCL: two ranks cannot share a GPU there).  The generated
    tokens must equal the un-split model's."""
    import torch.multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29900 + os.getpid() % 90
    procs = [ctxm.Process(target=_peer_worker, args=(r, 2, port, 3, 6, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    for p in procs:
        if p.is_alive():
            p.terminate()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) is True


@pytest.mark.parametrize("world", [2, 4])
def test_multi_process_pipeline_ingests_the_prompt_with_one_pass_per_stage(world):
    """Round 6 (config C3 composed with C4): two and four HipStages in as many processes on GPU 0; a 300-token prompt per sequence (one 256-token
    slab + a 44-token one) goes through PipelineDriver.prefill - thk_model_prefill_stage on every stage, the 300 x E rows through the
    mailbox's bulk slot, the last stage's pick back to rank 0 - and the ring continues from it.  Tokens must equal the un-split model's
    (full-model prefill + decode)."""
    import torch.multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29700 + os.getpid() % 90 + world
    procs = [ctxm.Process(target=_peer_worker, args=(r, world, port, 300, 6, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(400)
    for p in procs:
        if p.is_alive():
            p.terminate()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) is True
