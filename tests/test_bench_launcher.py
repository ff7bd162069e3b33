"""CPU tests of bench.py's own launcher: `--gpus N` must start N ranks or fail loudly, never measure
one GPU and call it N."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launch_command_and_env_for_world_2():
    import bench
    cmd = bench.launch_command(2, 29544, ["--gpus", "2", "--steps", "5"])
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29544"
    assert cmd[-4:] == ["--gpus", "2", "--steps", "5"] and cmd[-5].endswith("bench.py")
    env = bench.launch_env({"PATH": "/bin"})
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["MASTER_ADDR"] == "127.0.0.1" and env["PATH"] == "/bin"
    assert bench.launch_env({"HSA_ENABLE_IPC_MODE_LEGACY": "1"})["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"   # an explicit setting wins


def test_stage_cost_model_keeps_the_baseline_split():
    """bench.py's byte-based stage cost model (layer = its weight + KV bytes at 6.8 TB/s + 5 launches x 3.1 us; lm-head likewise) fed
    to pipeline.balanced_layer_split: for LLaMA-7B and 13B the lm-head costs about half a layer, so the uniform 32/16/8/4 split
    BASELINE.md names is the optimum at N = 2, 4, 8 - moving a layer off the last rank would make another rank the slowest."""
    import types
    import bench
    import __graft_entry__ as graft
    graft.load_package()
    from token_hawk_amd.pipeline import balanced_layer_split, layer_range, split_efficiency_bound
    for E, F, L in ((4096, 11008, 32), (5120, 13824, 40)):
        shape = types.SimpleNamespace(n_embd=E, n_ff=F, n_vocab=32000, n_layer=L)
        t_layer, t_head = bench.stage_cost_model_us(shape, 512)
        assert 60 < t_layer < 140 and 0.35 < t_head / t_layer < 0.7
        for N in (2, 4, 8):
            uni = [layer_range(L, r, N) for r in range(N)]
            assert balanced_layer_split(L, N, t_layer, t_head) == uni
            assert 0.85 < split_efficiency_bound(uni, t_layer, t_head) < 1.0


def run_bench(args, env_extra):
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_world_size_mismatch_is_an_error():
    r = run_bench(["--gpus", "8", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=1 but --gpus 8" in r.stderr and r.stdout.strip() == ""


def test_too_few_gpus_is_an_error_not_a_one_gpu_run():
    """On a box with fewer than N GPUs (this container has none) `bench.py --gpus 2` exits 2 and prints no JSON line."""
    r = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode == 2 and "refusing" in r.stderr and r.stdout.strip() == ""


def test_self_launch_world_2_runs_the_ring_protocol_on_cpu():
    """`bench.py --gpus 2` without torchrun starts its own two ranks; with --stub-stage (gloo, trivial CPU stage, no libthk) the
    whole N>1 flow runs here: rendezvous on 127.0.0.1, prime / steady / drain of the ring driver, barriers, MAX over ranks, ONE
    JSON line from rank 0 - so the first multi-GPU run is not the first time this code executes with N > 1."""
    import json
    r = run_bench(["--gpus", "2", "--stub-stage", "--steps", "4", "--warmup", "1", "--ctx", "8", "--master-port", str(29700 + os.getpid() % 200)], {})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["stub"] is True and d["n_gpus"] == 2 and d["ranks_joined"] == 2 and d["steps"] == 4 and d["scaling"] == "weak"
    assert d["items_per_rank_in_timed_region"] == 4 * 2          # every rank processed exactly steps * S items: no fill/drain inside
    assert "single_stream" in d and "steady ring" in d["timed_region"]
