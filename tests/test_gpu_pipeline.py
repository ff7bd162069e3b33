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
