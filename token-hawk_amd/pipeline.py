"""Layer-pipelined decode across the GPUs of one node (config C4).

Rank r owns layers [r*L/N, (r+1)*L/N); rank 0 also owns the embedding table, rank N-1 the
final norm + lm-head + greedy pick.  The only inter-rank traffic is the f32 hidden state
(E*4 bytes) from rank r to r+1 and the 4-byte token id from rank N-1 back to rank 0:
point-to-point over RCCL/xGMI (torch.distributed backend "nccl" == RCCL); no collective is
in the data path.  KV caches never move.

Schedule: S = N sequences are in flight.  Work item i = (step k = i // S, sequence s = i % S).
At micro-step j rank r processes item j - r, so every stage is busy on a different sequence and
the ring is synchronous: between micro-steps every rank posts ONE send (to r+1) and ONE receive
(from r-1) in a single grouped batch_isend_irecv, which cannot dead-lock even if sends are
rendezvous.  All device work is stream-ordered; the host never waits for the GPU inside the loop.

The stage object is duck-typed so the N>1 logic is covered on CPU with gloo (tests/
test_pipeline_gloo.py supplies an oracle-backed stage); the product stage is HipStage below.
"""
from __future__ import annotations

import os
import sys
import threading
import time
from dataclasses import dataclass, field

import torch
import torch.distributed as dist


def layer_range(n_layer: int, rank: int, world: int):
    """Contiguous split of the layers; the first n_layer % world ranks take one extra (32/16/8/4 per GPU)."""
    base, extra = divmod(n_layer, world)
    l0 = rank * base + min(rank, extra)
    return l0, l0 + base + (1 if rank < extra else 0)


def stage_cost(n_layers: int, first: bool, last: bool, t_layer: float, t_head: float, t_embed: float = 0.0) -> float:
    """Cost model of one stage: its layers, plus the embedding fetch on the first and final norm + lm-head + pick on the last."""
    return n_layers * t_layer + (t_embed if first else 0.0) + (t_head if last else 0.0)


def balanced_layer_split(n_layer: int, world: int, t_layer: float, t_head: float, t_embed: float = 0.0):
    """Contiguous layer counts per rank that minimise the slowest stage under stage_cost() - the ring runs at the pace of its
    slowest stage, and the last rank also carries the lm-head.  Exact (dynamic programme over (rank, layers used)); every rank
    keeps at least one layer; among the optimal splits the one closest to uniform wins, so the BASELINE split is returned
    whenever it is already optimal (LLaMA-7B: the lm-head costs 0.6 of a layer, so 4/4/.../4 is optimal at N = 8 - moving a
    layer off the last rank makes another rank slower than the head ever was).  Returns [(l0, l1)] * world."""
    assert 1 <= world <= n_layer
    INF = float("inf")
    uni = [b - a for a, b in (layer_range(n_layer, r, world) for r in range(world))]
    # best[r][k] = (max cost, distance from uniform) of ranks r.. when k layers remain for them
    best = [[(INF, INF)] * (n_layer + 1) for _ in range(world + 1)]
    pick = [[0] * (n_layer + 1) for _ in range(world + 1)]
    best[world][0] = (0.0, 0)
    for r in range(world - 1, -1, -1):
        for k in range(world - r, n_layer + 1):
            for n in range(1, k - (world - r - 1) + 1):
                rest = best[r + 1][k - n]
                if rest[0] == INF:
                    continue
                c = (max(stage_cost(n, r == 0, r == world - 1, t_layer, t_head, t_embed), rest[0]), abs(n - uni[r]) + rest[1])
                if c[0] < best[r][k][0] - 1e-12 or (abs(c[0] - best[r][k][0]) <= 1e-12 and c[1] < best[r][k][1]):
                    best[r][k] = c; pick[r][k] = n
    out, k, l0 = [], n_layer, 0
    for r in range(world):
        n = pick[r][k]
        out.append((l0, l0 + n)); l0 += n; k -= n
    return out


def split_efficiency_bound(split, t_layer: float, t_head: float, t_embed: float = 0.0) -> float:
    """Sum of the stage costs / (N x the slowest): what a zero-cost hand-off could reach relative to N perfectly balanced stages."""
    N = len(split)
    c = [stage_cost(b - a, r == 0, r == N - 1, t_layer, t_head, t_embed) for r, (a, b) in enumerate(split)]
    return sum(c) / (N * max(c))


class Watchdog:
    """`with Watchdog(seconds, what):` - if the body has not finished in time the process reports and exits with code 3.  The
    ring's device-side waits are stream-ordered (RCCL) or bounded on the device (thk_peer); what can still block for ever is the
    HOST waiting on a stream behind a hand-off whose peer died.  A hung rank would hang the node's other ranks and the driver with
    it; a dead one is an error the launcher reports."""

    def __init__(self, seconds: float, what: str, on_expire=None):
        self.seconds, self.what, self.on_expire = seconds, what, on_expire
        self._t = None

    def _fire(self):
        print(f"[pipeline watchdog] '{self.what}' did not finish within {self.seconds:.0f} s: a peer is gone or a hand-off is stuck; exiting (3)",
              file=sys.stderr, flush=True)
        if self.on_expire is not None:
            try:
                self.on_expire()
            except Exception:
                pass
        os._exit(3)

    def __enter__(self):
        if self.seconds and self.seconds > 0:
            self._t = threading.Timer(self.seconds, self._fire)
            self._t.daemon = True
            self._t.start()
        return self

    def __exit__(self, *a):
        if self._t is not None:
            self._t.cancel()
        return False


class _CudaView:
    """Zero-copy torch view of libthk device memory via __cuda_array_interface__."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class HipStage:
    """One pipeline stage on one MI355X, backed by thk_model_* (libthk.so)."""

    def __init__(self, thk, ctx, shape, rank: int, world: int, n_seq: int, device: torch.device, layers=None):
        l0, l1 = layers if layers is not None else layer_range(shape.n_layer, rank, world)      # layers: an uneven split (balanced_layer_split)
        self.l0, self.l1 = l0, l1
        flags = (thk.THK_STAGE_EMBED if rank == 0 else 0) | (thk.THK_STAGE_HEAD if rank == world - 1 else 0)
        self.model = thk.Model(ctx, shape, l0, l1, flags=flags, n_seq=n_seq)
        self.model.fill_synthetic()
        self.model.finalize()
        self.is_first, self.is_last = rank == 0, rank == world - 1
        E = shape.n_embd
        m = self.model
        self.hidden_in = [torch.as_tensor(_CudaView(m.hidden_in_ptr(s), (E,), "<f4"), device=device) for s in range(n_seq)]
        self.hidden_out = [torch.as_tensor(_CudaView(m.hidden_out_ptr(s), (E,), "<f4"), device=device) for s in range(n_seq)]
        self.token = [torch.as_tensor(_CudaView(m.token_dev_ptr(s), (1,), "<i4"), device=device) for s in range(n_seq)]
        # the rows a stage's prompt pass hands on (thk_model_prefill_stage works on them in place): one [n_ctx, E] f32 buffer per stage
        self._bulk_buf = thk.Buffer(ctx, shape.n_ctx * E * 4)
        self.bulk = torch.as_tensor(_CudaView(self._bulk_buf.ptr, (shape.n_ctx, E), "<f4"), device=device)
        self._thk = thk

    def prefill(self, seq: int, tokens, n_tokens: int, n_past: int):
        """This stage's layers over the prompt rows on the MFMA path (config C3 on a stage of C4): rows [0, n_tokens) of self.bulk are the
        input (token ids on the first stage) and become the output; the last stage leaves its greedy pick in the sequence's token slot.
        Every stage's position moves to n_past + n_tokens, the next slot to evaluate."""
        m, shape = self.model, self.model.shape
        m.prefill_stage(tokens if self.is_first else None, self._bulk_buf.ptr, n_tokens, n_past, seq=seq)
        if n_past + n_tokens < shape.n_ctx:
            m.seq_set(seq, 0, n_past + n_tokens)
        if self.is_last:                                 # llama_sample_top_p_top_k's temp <= 0 branch (th-llama.cpp:826-838) on the device, into the token slot
            ctx = m.ctx
            ctx.check(ctx.lib.thk_argmax(ctx.h, m.logits_dev_ptr(seq), shape.n_vocab, m.token_dev_ptr(seq)), "thk_argmax")

    def native_bulk(self, send_rows, recv_rows, nxt: int, prev: int):
        """The prompt rows over libthk's RCCL path: n_rows * E * 4 bytes - a bandwidth message (8 MB at 512 x 4096), not a latency one."""
        ctx, lib, E = self.model.ctx, self.model.ctx.lib, self.model.shape.n_embd
        if send_rows:
            ctx.check(lib.thk_pp_send(self.pp, self._bulk_buf.ptr, send_rows * E * 4, nxt), "thk_pp_send")
        if recv_rows:
            ctx.check(lib.thk_pp_recv(self.pp, self._bulk_buf.ptr, recv_rows * E * 4, prev), "thk_pp_recv")

    def peer_bulk(self, seq: int, send_rows, recv_rows):
        ctx, lib, E = self.model.ctx, self.model.ctx.lib, self.model.shape.n_embd
        if send_rows:
            ctx.check(lib.thk_peer_send_bulk(self.peer, seq, self._bulk_buf.ptr, send_rows * E * 4), "thk_peer_send_bulk")
        if recv_rows:
            ctx.check(lib.thk_peer_recv_bulk(self.peer, seq, self._bulk_buf.ptr, recv_rows * E * 4), "thk_peer_recv_bulk")

    def create_native_transport(self, rank: int, world: int, unique_id: bytes):
        """thk_pp_create (ncclCommInitRank inside: collective, may block) -> the handle; nothing of the stage is touched, so the call may run on a helper
        thread that the caller gives up on."""
        import ctypes as C
        ctx = self.model.ctx
        pp = C.c_void_p()
        ctx.check(ctx.lib.thk_pp_create(ctx.h, world, rank, unique_id, C.byref(pp)), "thk_pp_create")
        return pp

    def attach_native_transport(self, rank: int, world: int, unique_id: bytes):
        """Use libthk's own RCCL point-to-point path (thk_pp_*) instead of torch.distributed P2P ops."""
        self.pp = self.create_native_transport(rank, world, unique_id)

    def native_exchange(self, sends, recvs):
        """sends/recvs: lists of (kind, seq, peer), kind in {'hidden','token'}; one grouped RCCL step."""
        ctx, lib, pp, h = self.model.ctx, self.model.ctx.lib, self.pp, self.model.h
        ctx.check(lib.thk_pp_group_begin(pp), "thk_pp_group_begin")
        for kind, seq, peer in sends:
            ctx.check((lib.thk_pp_send_hidden if kind == "hidden" else lib.thk_pp_send_token)(pp, h, seq, peer), "thk_pp_send")
        for kind, seq, peer in recvs:
            ctx.check((lib.thk_pp_recv_hidden if kind == "hidden" else lib.thk_pp_recv_token)(pp, h, seq, peer), "thk_pp_recv")
        ctx.check(lib.thk_pp_group_end(pp), "thk_pp_group_end")

    def attach_peer_transport(self, n_seq: int):
        """The mailbox transport (thk_peer_*, no communication library).  Returns this stage's 64-byte IPC handle; hand it to the
        PREVIOUS stage, then call connect_peer() with the handle of the NEXT stage (None: single-stage ring)."""
        import ctypes as C
        ctx = self.model.ctx
        peer = C.c_void_p()
        ctx.check(ctx.lib.thk_peer_create(ctx.h, self.model.h, n_seq, C.byref(peer)), "thk_peer_create")
        self.peer = peer
        buf = C.create_string_buffer(64)
        ctx.check(ctx.lib.thk_peer_export(peer, buf), "thk_peer_export")
        return bytes(buf.raw)

    def connect_peer(self, next_handle):
        ctx = self.model.ctx
        ctx.check(ctx.lib.thk_peer_connect(self.peer, next_handle), "thk_peer_connect")

    def peer_exchange(self, sends, recvs):
        """sends/recvs: lists of (kind, seq); every send is enqueued before any wait (a single-stage ring waits for itself)."""
        ctx, lib = self.model.ctx, self.model.ctx.lib
        for kind, seq in sends:
            ctx.check(lib.thk_peer_send(self.peer, seq, 0 if kind == "hidden" else 1), "thk_peer_send")
        for kind, seq in recvs:
            ctx.check(lib.thk_peer_recv(self.peer, seq, 0 if kind == "hidden" else 1), "thk_peer_recv")

    def peer_check(self):
        self.model.ctx.check(self.model.ctx.lib.thk_peer_check(self.peer), "thk_peer_check")

    def set_seq(self, seq: int, token: int, pos: int):
        self.model.seq_set(seq, token, pos)

    def set_token(self, seq: int, token: int):
        self.model.seq_set_token(seq, int(token))

    def step(self, seq: int, advance: bool):
        self.model.decode_step(seq, advance)

    def generated(self, seq: int):
        return self.model.seq_get(seq)[0].tolist()


@dataclass
class PipelineResult:
    items: int          # work items this rank processed
    micro_steps: int
    topped_up: int = 0  # drain(): extra items issued so that every sequence ends on the same step


@dataclass
class HandoffReport:
    ok: bool
    checked: int                      # payloads this rank verified
    handoff_us: float                 # average time of one ring hand-off (all ranks sending and receiving at once), no compute
    errors: list = field(default_factory=list)


class PipelineDriver:
    """Synchronous ring over an endless stream of work items: item i = (step i // S, sequence i % S); at global micro-step j rank
    r processes item j - r and then exchanges (one send, one receive, grouped).  Two ways to drive it:

    * run(steps)                     self-contained: fills the ring, processes steps * S items on every rank, drains it
                                     (steps * S + N - 1 micro-steps).  Used for the prompt and by the tests.
    * prime() / steady(steps) / drain()   the ring stays full between calls: steady(steps) is exactly steps * S micro-steps in
                                     which EVERY rank processes one item, so a timed region around it holds no fill or drain
                                     (round 2 timed run(): 7 of 167 micro-steps at N = 8, steps = 20 were fill/drain).
    """

    def __init__(self, stage, rank: int, world: int, n_seq: int, force_ring: bool = False):
        assert n_seq == world or (world == 1 and not force_ring), "the synchronous ring schedule needs one sequence per stage"
        self.stage, self.rank, self.world, self.S = stage, rank, world, n_seq
        self.force_ring = force_ring      # world == 1 only: still post the (self) send/recv pair, to exercise the P2P plumbing
        self.prev, self.next = (rank - 1) % world, (rank + 1) % world
        self.base = 0                     # items [0, base) have left the last stage
        self.j = 0                        # next global micro-step
        self.primed = False

    @property
    def ring(self) -> bool:
        return self.world > 1 or self.force_ring

    def _xfer(self, send_seq, recv_seq):
        """One grouped hand-off on the stage's transport: send this stage's output of sequence send_seq (hidden state, or the token
        from the last stage) to the next rank, receive the input of sequence recv_seq from the previous one (None = nothing)."""
        st = self.stage
        skind, rkind = ("token" if st.is_last else "hidden"), ("token" if st.is_first else "hidden")
        if send_seq is None and recv_seq is None:
            return
        if getattr(st, "peer", None) is not None:        # mailbox transport (thk_peer_*): stores into the next stage's memory, no library
            st.peer_exchange([(skind, send_seq)] if send_seq is not None else [], [(rkind, recv_seq)] if recv_seq is not None else [])
            return
        if getattr(st, "pp", None) is not None:          # native RCCL transport (thk_pp_*), same schedule
            st.native_exchange([(skind, send_seq, self.next)] if send_seq is not None else [], [(rkind, recv_seq, self.prev)] if recv_seq is not None else [])
            return
        ops = []
        if send_seq is not None:                         # the token feeds item i_done + S on rank 0
            ops.append(dist.P2POp(dist.isend, st.token[send_seq] if st.is_last else st.hidden_out[send_seq], self.next))
        if recv_seq is not None:
            ops.append(dist.P2POp(dist.irecv, st.token[recv_seq] if st.is_first else st.hidden_in[recv_seq], self.prev))
        for w in dist.batch_isend_irecv(ops):
            w.wait()                                     # stream-ordered for NCCL; blocking (bounded by the group's timeout) for gloo

    def _exchange(self, j: int, lo: int, hi):
        """Grouped P2P after micro-step j: send item (j - rank)'s output, receive the input of item (j + 1 - rank).
        Only items in [lo, hi) exist (hi = None: no upper bound, the ring is kept full)."""
        r, N, S = self.rank, self.world, self.S
        live = (lambda i: lo <= i and (hi is None or i < hi))
        i_done = j - r                                   # item this rank just finished
        i_prev = j - ((r - 1) % N)                       # item the previous rank just finished
        self._xfer(i_done % S if live(i_done) else None, i_prev % S if live(i_prev) else None)

    def validate_handoff(self, reps: int = 16, sync=None, fence=None) -> HandoffReport:
        """Before anything is timed: every rank writes a known pattern into each sequence's output slot (hidden state, or token on
        the last stage), the ring hands all of them over on the transport in use, and every receiver checks what arrived against the
        pattern its predecessor must have written.  Then `reps` bare hand-offs per sequence are timed (every rank sends and receives in
        each, as in a steady micro-step) -> handoff_us.  Clobbers hidden_in / token: call it before the sequences are set up.
        sync: callable that waits for the device (None: CPU stage).  All ranks must call it together.
        fence: callable that is a CROSS-RANK barrier (e.g. an all-reduce on the control plane).  The mailbox transport has no
        back-pressure of its own - a (sequence, kind) slot may only be rewritten after its reader has consumed it, which the ring
        schedule guarantees by its S-in-flight flow control and this loop does not - so every round of the check and every timed
        repetition ends with sync + fence: a rank a few ms ahead can then never overwrite a payload its successor is still checking
        (a healthy transport would report a false failure, or the timing loop would copy torn payloads).  Timed: the hand-offs
        only, not the fences.  None with more than one rank: torch.distributed.barrier() when a process group exists."""
        st, r, N, S = self.stage, self.rank, self.world, self.S
        if not self.ring:
            return HandoffReport(True, 0, 0.0)
        E = st.hidden_out[0].numel()

        def pattern(rank, s, rep):                       # exactly representable f32 values, different for every (rank, sequence, element, round)
            base = torch.arange(E, dtype=torch.float32)
            return (base * 0.5 + float(1000 * rank + 10 * s + rep)) * (-1.0 if (rank + s) % 2 else 1.0)

        def token_code(rank, s, rep):
            return 7 + 100 * rank + 10 * s + rep

        if fence is None and self.world > 1:
            fence = dist.barrier if (dist is not None and dist.is_available() and dist.is_initialized()) else None

        def settle():                                    # everything this rank sent has landed AND every rank has finished reading
            if sync is not None:
                sync()
            if fence is not None:
                fence()

        errors, checked = [], 0
        for rep in range(2):                             # twice: the second round proves nothing stale from the first is read
            for s in range(S):
                if st.is_last:
                    st.token[s].copy_(torch.tensor([token_code(r, s, rep)], dtype=torch.int32))
                else:
                    st.hidden_out[s].copy_(pattern(r, s, rep))
            for s in range(S):
                self._xfer(s, s)
            if sync is not None:
                sync()
            for s in range(S):
                if st.is_first:
                    got, want = int(st.token[s].cpu()[0]), token_code(self.prev, s, rep)
                    if got != want:
                        errors.append(f"rank {r} seq {s} round {rep}: token {got} != {want} from rank {self.prev}")
                else:
                    got, want = st.hidden_in[s].cpu(), pattern(self.prev, s, rep)
                    if not torch.equal(got, want):
                        bad = int((got != want).sum())
                        errors.append(f"rank {r} seq {s} round {rep}: {bad} of {E} hidden-state words differ from rank {self.prev}'s pattern")
                checked += 1
            settle()                                     # nobody starts round 2 (or the timing) while a neighbour still checks round 1
        spent = 0.0
        for k in range(reps):
            t0 = time.perf_counter()
            for s in range(S):
                self._xfer(s, s)
            if sync is not None:
                sync()
            spent += time.perf_counter() - t0
            if fence is not None:
                fence()                                  # a slot is rewritten only after every rank has taken the previous payload
        us = spent / max(1, reps * S) * 1e6
        return HandoffReport(not errors, checked, us, errors)

    def _bulk(self, seq: int, send_rows: int, recv_rows: int):
        """The prompt rows of one sequence to the next stage / from the previous one, on the transport in use."""
        st = self.stage
        if getattr(st, "peer", None) is not None:
            st.peer_bulk(seq, send_rows, recv_rows)
        elif getattr(st, "pp", None) is not None:
            st.native_bulk(send_rows, recv_rows, self.next, self.prev)
        else:
            ops = []
            if send_rows:
                ops.append(dist.P2POp(dist.isend, st.bulk[:send_rows], self.next))
            if recv_rows:
                ops.append(dist.P2POp(dist.irecv, st.bulk[:recv_rows], self.prev))
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def prefill(self, prompts, n_past: int = 0, feed_back: bool = True) -> PipelineResult:
        """Prompt ingestion with ONE batched pass per stage and sequence (round 6; config C3 composed with C4) instead of len(prompts) ring
        revolutions: prompts[:, s] (int ids, [M, S]; only the first stage reads them) goes through stage 0's layers as one MFMA prompt pass,
        its M x E output rows travel to stage 1 as one bandwidth message, and so on - stage r works on sequence s while stage r + 1 works
        on s - 1.  The row hand-offs only point forward (a chain, not a ring), so nothing can dead-lock and no grouping is needed; the last
        stage's greedy picks go back to the first stage's token slots after the last forward message (feed_back), exactly what the ring
        leaves there after run(M, forced_tokens=prompts), so run() / prime() continue from them.  The ring must be empty.  Matches the reference's batch branch,
        which is per layer (th-llama.cpp:305-311, :365-404)."""
        st, S, N = self.stage, self.S, self.world
        assert not self.primed, "prefill() needs an empty ring: drain() first"
        M = len(prompts)
        for s in range(S):
            if not st.is_first:
                self._bulk(s, 0, M)
            st.prefill(s, [int(t) for t in prompts[:, s]] if st.is_first else None, M, n_past)
            if not st.is_last:
                self._bulk(s, M, 0)
        # the picks travel back only after EVERY forward message of this rank is behind it: a send completes when its receive is posted (RCCL
        # kernels and gloo alike), so a token send between two bulk receives would wait for rank 0, which waits for its next bulk send
        if feed_back and N > 1:
            for s in range(S):
                if st.is_last:
                    self._xfer(s, None)
                elif st.is_first:
                    self._xfer(None, s)
        return PipelineResult(S, S)

    def _micro(self, n_micro: int, lo: int, hi, advance: bool, forced_tokens=None) -> int:
        """n_micro global micro-steps from self.j on; returns the number of items this rank processed."""
        st, r, S = self.stage, self.rank, self.S
        done = 0
        for j in range(self.j, self.j + n_micro):
            i = j - r
            if lo <= i and (hi is None or i < hi):
                if st.is_first and forced_tokens is not None:
                    k, s = divmod(i - lo, S)             # lo is a multiple of S here (checked by run())
                    st.set_token(s, forced_tokens[k][s])
                st.step(i % S, advance)
                done += 1
            self._exchange(j, lo, hi)
        self.j += n_micro
        return done

    def run(self, steps: int, advance: bool, forced_tokens=None) -> PipelineResult:
        """Advance every sequence by `steps` tokens, ring empty before and after.  forced_tokens[k][s] (rank 0 only) overrides
        the fed-back token with a prompt token (prefill through the same path)."""
        st, S = self.stage, self.S
        assert not self.primed, "run() needs an empty ring: drain() first"
        total = steps * S
        if not self.ring:
            for i in range(total):
                k, s = divmod(i, S)
                if forced_tokens is not None:
                    st.set_token(s, forced_tokens[k][s])
                st.step(s, advance)
            return PipelineResult(total, total)
        assert forced_tokens is None or self.base % S == 0, "forced tokens need the ring at a step boundary"
        lo = self.base
        n_micro = total + self.world - 1
        done = self._micro(n_micro, lo, lo + total, advance, forced_tokens)
        self.base = lo + total
        self.j = self.base                                # an empty ring restarts at micro-step == first item
        return PipelineResult(done, n_micro)

    def prime(self, advance: bool) -> PipelineResult:
        """Fill the ring: N - 1 micro-steps, after which rank r has item base + N - 2 - r behind it and every later
        micro-step finds work on every rank."""
        assert not self.primed
        self.primed = True
        if not self.ring:
            return PipelineResult(0, 0)
        n = self.world - 1
        return PipelineResult(self._micro(n, self.base, None, advance), n)

    def steady(self, steps: int, advance: bool) -> PipelineResult:
        """steps * S micro-steps with the ring full: every rank processes exactly steps * S items."""
        assert self.primed, "prime() first"
        total = steps * self.S
        if not self.ring:
            for i in range(total):
                self.stage.step(i % self.S, advance)
            return PipelineResult(total, total)
        return PipelineResult(self._micro(total, self.base, None, advance), total)

    def drain(self, advance: bool, align: bool = True) -> PipelineResult:
        """Let the items already issued by rank 0 leave the last stage; the ring is empty afterwards.  prime() issued N - 1 items
        beyond whole steps, so with align (default) rank 0 first issues the S - (N - 1) % S items that complete the step: every
        sequence then ends the same number of tokens ahead and the ring is back on a step boundary (run(forced_tokens=...) needs
        one).  align=False keeps round 3's behaviour (sequences 0..N-2 end one token ahead of the rest)."""
        assert self.primed
        self.primed = False
        if not self.ring:
            return PipelineResult(0, 0)
        hi = self.j                                       # rank 0 has issued items [base, j)
        top = (-(hi - self.base)) % self.S if align else 0
        hi += top
        n = top + self.world - 1
        done = self._micro(n, self.base, hi, advance)
        self.base = hi
        self.j = hi
        return PipelineResult(done, n, top)
