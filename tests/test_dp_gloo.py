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


def _overlap(a, b):
    lo_a, lo_b = a.data_ptr(), b.data_ptr()
    return lo_a < lo_b + b.numel() * b.element_size() and lo_b < lo_a + a.numel() * a.element_size()


def _worker_world8(rank, world, port, q, case):
    """One rank of the world-size-8 arithmetic checks (the target job: one node of 8 MI355X).  case = (mode, bucket elements,
    branch): branch "gloo" = GradReducer's gloo emulation, "pg" = its torch.distributed reduce_scatter_tensor /
    all_gather_into_tensor branch (what runs over RCCL), exercised here through stand-ins built on gloo's all_reduce /
    all_gather that also REFUSE an output aliasing the input."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from pdm.training.bilevel import GradReducer
        mode, bucket, branch = case
        N = 1003                                              # prime-ish arena: ragged head in every configuration

        class Store:
            total = N
            master = torch.zeros(N)
            grad = (torch.arange(N, dtype=torch.float32) % 97 + 1) * (rank + 1)
        red = GradReducer(Store, bucket_mb=1, mode=mode)
        red.bucket = bucket
        calls = {"rs": 0, "ag": 0}
        if branch == "pg":
            red.backend = "nccl"                              # take the reduce_scatter_tensor / all_gather_into_tensor branch

            def rs(out, inp, op=None):
                assert not _overlap(out, inp), "reduce_scatter_tensor output aliases its input"
                assert inp.numel() == world * out.numel()
                tmp = inp.clone()
                dist.all_reduce(tmp)
                out.copy_(tmp[rank * out.numel():(rank + 1) * out.numel()])
                calls["rs"] += 1

            def ag(out, inp):
                assert not _overlap(out, inp), "all_gather_into_tensor output aliases its input"
                assert out.numel() == world * inp.numel()
                parts = [torch.empty_like(inp) for _ in range(world)]
                dist.all_gather(parts, inp.contiguous())
                out.copy_(torch.cat(parts))
                calls["ag"] += 1
            dist.reduce_scatter_tensor, dist.all_gather_into_tensor = rs, ag
        red.begin()
        launched = []
        orig = red._launch
        red._launch = lambda lo, hi: (launched.append((lo, hi)), orig(lo, hi))[1]
        for lo in (900, 512, 511, 64, 0):                     # the backward tape reports these offsets as final, in order
            red.ready_down_to(lo)
        scale = red.finish()
        # every element reduced exactly once, buckets contiguous from the tail, nothing left
        assert launched and launched[0][1] == N and launched[-1][0] == 0, launched
        assert all(a[0] == b[1] for a, b in zip(launched, launched[1:])), launched
        assert all(hi - lo == bucket for lo, hi in launched[:-1]) and launched[-1][1] - launched[-1][0] <= max(bucket, N), launched
        expect_coll = 0
        for lo, hi in launched:
            n = hi - lo
            if mode == "allreduce":
                expect_coll += 1
            else:
                expect_coll += (2 if n - n % world else 0) + (1 if n % world else 0)
        assert red.n_collectives == expect_coll, (red.n_collectives, expect_coll, launched)
        if branch == "pg":
            bodies = sum(1 for lo, hi in launched if (hi - lo) - (hi - lo) % world)
            assert calls == {"rs": bodies, "ag": bodies}, (calls, bodies)
        expect = (torch.arange(N, dtype=torch.float32) % 97 + 1) * sum(r + 1 for r in range(world))
        q.put((rank, bool(torch.equal(Store.grad, expect)) and scale == 1.0 / world, case))
        dist.destroy_process_group()
    except Exception as e:
        q.put((rank, False, repr(e)))
        raise


def _run_world8(case, port_base):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = port_base + os.getpid() % 1000
    ps = [ctx.Process(target=_worker_world8, args=(r, 8, port, q, case)) for r in range(8)]
    [p.start() for p in ps]
    res = [q.get(timeout=180) for _ in ps]
    [p.join(60) for p in ps]
    assert all(ok for _, ok, _ in res), res
    assert all(p.exitcode == 0 for p in ps)


def test_bucket_share_tail_arithmetic_world8_gloo():
    """World size 8 (the job the metric is quoted on): tail-first buckets with a ragged head, for an odd bucket (255: share
    body 248 + tail 7), a bucket SMALLER than the world (5: no share body at all, the tail all-reduce carries it), a bucket
    the world divides (64: no tail) and one bucket larger than the arena; both exchange modes; sums, bucket order, contiguity
    and the number of collectives."""
    for i, case in enumerate([("rs_ag", 255, "gloo"), ("rs_ag", 5, "gloo"), ("rs_ag", 64, "gloo"), ("allreduce", 255, "gloo"),
                              ("rs_ag", 4096, "gloo")]):
        _run_world8(case, 31512 + 7 * i)


def test_rs_ag_process_group_branch_never_aliases_world8():
    """The branch that runs over RCCL (dist.reduce_scatter_tensor / all_gather_into_tensor): its share is a scratch buffer, never
    a slice of the bucket - checked with stand-ins that refuse aliased arguments - and the sums equal the all-reduce's."""
    for i, case in enumerate([("rs_ag", 255, "pg"), ("rs_ag", 64, "pg")]):
        _run_world8(case, 32512 + 7 * i)


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
