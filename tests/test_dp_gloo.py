"""CPU, world_size 2 over gloo: the data-parallel gradient path (bucketed tail-first all-reduce + mean folded into the
optimiser scale) and LR-schedule scaling, i.e. the N>1 logic of bench.py / the trainer without a GPU."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception as e:      # report instead of leaving the parent waiting on the queue
        q.put((rank, False, repr(e)))
        raise


def _worker_body(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pdm.training.bilevel import GradReducer

    class Store:
        total = 1000
        master = torch.zeros(1000)
        grad = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    red = GradReducer(Store, bucket_mb=1)
    red.bucket = 256                      # elements: forces several buckets + a ragged head
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
