"""CPU, world_size 2 over gloo: the data-parallel gradient path (bucketed tail-first all-reduce + mean folded into the
optimiser scale) and LR-schedule scaling, i.e. the N>1 logic of bench.py / the trainer without a GPU."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q, mode="allreduce"):
    try:
        _worker_body(rank, world, port, q, mode)
    except Exception as e:      # report instead of leaving the parent waiting on the queue
        q.put((rank, False, repr(e)))
        raise


def _worker_body(rank, world, port, q, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pdm.training.bilevel import GradReducer

    class Store:
        total = 1000
        master = torch.zeros(1000)
        grad = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    red = GradReducer(Store, bucket_mb=1, mode=mode)
    red.bucket = 256 if mode == "allreduce" else 255     # elements: several buckets + a ragged head (rs_ag: 255 is odd, so
    if mode == "rs_ag":                                  # every bucket has a share body AND a one-element tail all-reduce)
        _rs_ag_body(red, Store, rank, world, q)
        return
    red.begin()
    launched = []
    orig = red._launch
    red._launch = lambda lo, hi: (launched.append((lo, hi)), orig(lo, hi))[1]
    red.ready_down_to(700)                # backward reached arena offset 700: one whole bucket [744,1000) can go
    assert launched == [(744, 1000)], launched
    red.ready_down_to(300)                # one more whole bucket fits above offset 300
    assert launched == [(744, 1000), (488, 744)], launched
    scale = red.finish()                  # the ragged head
    assert launched[-1] == (0, 488) and scale == 1.0 / world
    expect = torch.arange(1000, dtype=torch.float32) * sum(r + 1 for r in range(world))
    q.put((rank, bool(torch.equal(Store.grad, expect)), scale))
    dist.destroy_process_group()


def _rs_ag_body(red, Store, rank, world, q):
    red.begin()
    launched = []
    orig = red._launch
    red._launch = lambda lo, hi: (launched.append((lo, hi)), orig(lo, hi))[1]
    red.ready_down_to(700)
    assert launched == [(745, 1000)], launched
    assert red.n_collectives == 3, red.n_collectives      # reduce-scatter + all-gather of 254 elements, all-reduce of 1
    red.ready_down_to(0)
    assert launched == [(745, 1000), (490, 745), (235, 490)], launched
    scale = red.finish()                                  # the head [0, 235): 234 in shares + 1
    assert launched[-1] == (0, 235) and scale == 1.0 / world
    expect = torch.arange(1000, dtype=torch.float32) * sum(r + 1 for r in range(world))
    q.put((rank, bool(torch.equal(Store.grad, expect)), scale))
    dist.destroy_process_group()


def test_bucketed_reduce_scatter_allgather_world2_gloo():
    """PDMK_DP_MODE=rs_ag: every bucket as reduce-scatter + all-gather of `world` equal shares plus an all-reduce of the
    (n mod world) tail - the bucket / share arithmetic of SURVEY 5 / 8e, over gloo (share-wise reduce + broadcast, the same
    data movement; over RCCL: ncclReduceScatter / ncclAllGather in place)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30512 + os.getpid() % 1000
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, "rs_ag")) for r in range(2)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in ps]
    [p.join(60) for p in ps]
    assert all(ok for _, ok, _ in res), res
    assert all(p.exitcode == 0 for p in ps)


def test_bucketed_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29512 + os.getpid() % 1000
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in ps]
    [p.join(60) for p in ps]
    assert all(ok for _, ok, _ in res), res
    assert all(abs(s - 0.5) < 1e-12 for _, _, s in res), res
    assert all(p.exitcode == 0 for p in ps)


def test_bench_self_launch_sets_the_rank_environment(tmp_path, monkeypatch):
    """`python bench.py --gpus N` from a bare shell: the parent starts N children with the torchrun environment, never
    touches the GPU itself and returns the worst exit code.  Here the child is replaced by a stub that records its
    environment (no GPU in this container), which is exactly what the launcher hands to the real ranks."""
    import importlib.util
    import subprocess
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    stub = tmp_path / "stub.py"
    stub.write_text("import os, sys\n"
                    "open(os.path.join(os.environ['STUB_DIR'], 'r' + os.environ['RANK']), 'w').write(' '.join(\n"
                    "    os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')) + '|' + ' '.join(sys.argv[1:]))\n"
                    "sys.exit(3 if os.environ['RANK'] == '1' and os.environ.get('STUB_FAIL') else 0)\n")
    real_popen = subprocess.Popen
    monkeypatch.setattr(subprocess, "Popen", lambda cmd, env=None, **kw: real_popen([cmd[0], str(stub)] + cmd[2:], env=env, **kw))
    monkeypatch.setenv("STUB_DIR", str(tmp_path))
    monkeypatch.setenv("PDMK_BENCH_REHEARSAL", "1")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    assert bench.launch_ranks(types.SimpleNamespace(gpus=2)) == 0
    recs = [(tmp_path / f"r{r}").read_text() for r in range(2)]
    envs = [r.split("|")[0].split() for r in recs]
    assert [e[0] for e in envs] == ["0", "1"] and all(e[2] == "2" and e[3] == "127.0.0.1" for e in envs)
    assert envs[0][4] == envs[1][4] and all(r.endswith("--gpus 2 --steps 3") for r in recs)
    monkeypatch.setenv("STUB_FAIL", "1")
    assert bench.launch_ranks(types.SimpleNamespace(gpus=2)) == 3
    # without rehearsal mode the launcher refuses to start more ranks than there are GPUs (none in this container)
    monkeypatch.delenv("PDMK_BENCH_REHEARSAL")
    import pytest
    with pytest.raises(SystemExit):
        bench.launch_ranks(types.SimpleNamespace(gpus=2))
