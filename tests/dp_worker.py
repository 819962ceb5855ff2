"""One data-parallel rank of tests/test_dp_equivalence_gpu.py (not a test module).

Rehearsal of the N-rank path on a one-GPU box: every rank sits on cuda:0 and the process group is gloo (the product
path is the same code over RCCL).  The global batch of 4 samples is split row-wise over `world` ranks; every rank runs
3 bilevel iterations (main step + AdamW each, upper step + its AdamW on the third) of the tiny fp32 engine, eager from the
backward tape or as segmented hipGraph replay, and rank 0 saves:
  grad   the rank-mean gradient arena after the first main step's backward + all-reduce (before any optimiser step)
  master the fp32 master arena after the 3 iterations
Reference: DDP wrap + accelerator.backward, pdm/training/trainer.py:117-129, 2719-2724, 2782, 2808.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (os.path.join(ROOT, "unlearn-ft_amd"), os.path.join(ROOT, "oracle")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

GLOBAL_B, ITERS, UPPER_AT = 4, 3, 2
LR, UPPER_LR = 1e-4, 2e-4


def DEV_INDEX():
    """LOCAL_RANK when the box has a device for it, else 0 (all ranks share the one GPU)."""
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return local if local < torch.cuda.device_count() else 0


def DEV():
    return f"cuda:{DEV_INDEX()}"


def global_batches():
    g = torch.Generator().manual_seed(97)
    out = []
    for _ in range(ITERS + 1):
        out.append(dict(lat=torch.randn(GLOBAL_B, 4, 16, 16, generator=g), noise=torch.randn(GLOBAL_B, 4, 16, 16, generator=g),
                        t=torch.randint(0, 1000, (GLOBAL_B,), generator=g), ehs=torch.randn(GLOBAL_B, 13, 64, generator=g),
                        empty=torch.randn(1, 13, 64, generator=g).expand(GLOBAL_B, 13, 64).contiguous()))
    return out


def run(mode, world, rank, out=None):
    from pdm_ref import arch as oarch, weights as oweights
    from pdm_ref.config import UNetConfig as OCfg
    from pdm import _pdmk as k
    from pdm.models.unet.spec import UNetConfig
    from pdm.models.unet.unet_2d_conditional import UNet2DConditionModelPruned
    from pdm.training.bilevel import BilevelStepper, GraphedBilevel
    ocfg, cfg = OCfg.tiny(), UNetConfig.tiny()
    dense = oweights.init_dense_state_dict(ocfg, seed=0)
    av = oarch.random_arch_vector(ocfg, 0.55, seed=0, drop_depth=(1, 9))
    student = UNet2DConditionModelPruned(cfg, av, DEV(), torch.float32, train=True, init=False)
    student.load_dense_or_pruned(dense)
    teacher = UNet2DConditionModelPruned(cfg, None, DEV(), torch.float32, train=False, init=False)
    teacher.load_dense_or_pruned(dense)
    st = BilevelStepper(student, teacher, lr=LR, upper_lr=UPPER_LR, bilevel=True, bucket_mb=1)
    st.reducer.bucket = max(1024, student.store.total // 5)        # several buckets + a ragged head on the tiny arena
    assert st.world == world
    B = GLOBAL_B // world
    rows = slice(rank * B, (rank + 1) * B)
    data = [{n: v[rows].to(DEV()) for n, v in d.items()} for d in global_batches()]
    store = student.store
    init = store.master.clone()
    graphs = None
    prefetch = mode == "graph-prefetch"     # the teacher pass of the next step queued behind this step's loss heads, on every rank
    if mode in ("graph", "graph-prefetch"):
        graphs = GraphedBilevel(st, B, 4, 16, 16, 13, 64, segments=3, stream_opt=True, prefetch=prefetch)
        graphs.force_segments = True            # world 1 takes the multi-graph replay path too
        graphs.capture(bilevel=True)
        assert len(graphs.g_main.bwd) == 3 and graphs.g_main.teacher is not None      # fp32 engine: no lockstep
        assert torch.equal(store.master, init)  # capture restored the training state
    # ---- gradient of the first main step, reduced over the ranks, no optimiser
    d = data[ITERS]
    if graphs is None:
        st.main_step(d["lat"], d["noise"], d["t"], d["ehs"])
        scale = st._gscale
    else:
        graphs._load(d["lat"], d["noise"], d["t"], d["ehs"])
        graphs._claim(False, None, (d["lat"], d["noise"], d["t"], d["ehs"]))      # (prefetch mode: the teacher graph's own inputs)
        graphs._replay_step(graphs.g_main, None)
        scale = 1.0 / world
    torch.cuda.synchronize()
    grad = (store.grad * scale).cpu()
    k.zero_(store.grad)
    # ---- 3 bilevel iterations
    for it in range(ITERS):
        d = data[it]
        if graphs is None:
            st.main_step(d["lat"], d["noise"], d["t"], d["ehs"])
            st.optimizer_step(upper=False)
            if it == UPPER_AT:
                st.upper_step(d["lat"], d["noise"], d["t"], d["ehs"], d["empty"])
                st.optimizer_step(upper=True)
        elif not prefetch:
            graphs.main(d["lat"], d["noise"], d["t"], d["ehs"])
            if it == UPPER_AT:
                graphs.upper(d["lat"], d["noise"], d["t"], d["ehs"], d["empty"])
        else:
            tup = lambda q: (q["lat"], q["noise"], q["t"], q["ehs"])
            nxt = dict(next_batch=tup(data[it + 1]), next_id=it + 1) if it + 1 < ITERS else {}
            if it == UPPER_AT:
                graphs.main(*tup(d), batch_id=it, next_upper=tup(d) + (d["empty"],), upper_id=("u", it))
                graphs.upper(*tup(d), d["empty"], batch_id=("u", it), **nxt)
            else:
                graphs.main(*tup(d), batch_id=it, **nxt)
    torch.cuda.synchronize()
    if prefetch:
        assert graphs.prefetch and graphs.prefetch_hits == ITERS, graphs.prefetch_hits      # main steps 1 .. ITERS - 1 and the upper step
    res = {"grad": grad, "master": store.master.cpu(), "init": init.cpu(), "lr": st.opt.current_lr()}
    if out and rank == 0:
        torch.save(res, out)
    return res


def run_comm(world, rank, out):
    """RCCL with world > 1, one device per rank (needs >= world GPUs): the library communicator's reduce-scatter + all-gather
    (in place, RCCL's documented placement) and GradReducer's torch.distributed rs_ag branch (scratch share) against a plain
    all-reduce of the same buckets - bit-equal is not promised by RCCL across algorithms, so sums of small integers are used,
    which every summation order gives exactly."""
    from pdm import _pdmk as k
    from pdm.training.bilevel import GradReducer
    dev = torch.device(DEV())
    n = 8 * 1031
    x = ((torch.arange(n, device=dev) % 251) + 1).float() * (rank + 1)
    want = ((torch.arange(n, device=dev) % 251) + 1).float() * sum(r + 1 for r in range(world))
    box = [k.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    comm = k.Comm(box[0], rank, world)
    assert (comm.world_size(), comm.rank_id()) == (world, rank)
    a = x.clone()
    comm.all_reduce_sum_(a)
    m = n - n % world
    b = x[:m].clone()
    comm.reduce_scatter_sum_(b)
    per = m // world
    torch.cuda.synchronize()
    assert torch.equal(b[rank * per:(rank + 1) * per], want[rank * per:(rank + 1) * per])
    comm.all_gather_(b)
    torch.cuda.synchronize()
    assert torch.equal(a, want) and torch.equal(b, want[:m])
    comm.close()
    for mode, native in (("rs_ag", False), ("rs_ag", True), ("allreduce", True)):
        os.environ.pop("PDMK_COMM", None)
        if native:
            os.environ["PDMK_COMM"] = "native"

        class Store:
            total = n
            master = torch.zeros(n, device=dev)
            grad = x.clone()
        red = GradReducer(Store, bucket_mb=1, mode=mode)
        red.bucket = 1001
        red.begin()
        red.ready_down_to(4000)
        red.ready_down_to(0)
        assert red.finish() == 1.0 / world
        torch.cuda.synchronize()
        assert torch.equal(Store.grad, want), (mode, native)
    if out and rank == 0:
        torch.save({"ok": True}, out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", required=True, choices=["eager", "graph", "graph-prefetch", "comm"])
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    if a.mode == "comm":
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
        dist.init_process_group("nccl", rank=rank, world_size=world)
        run_comm(world, rank, a.out)
        dist.barrier()
        dist.destroy_process_group()
        return
    # one device per rank when the box has them (LOCAL_RANK -> set_device, as bench.py / Trainer do); on a one-GPU box both
    # ranks share cuda:0 (RCCL refuses that, gloo does not)
    torch.cuda.set_device(DEV_INDEX())
    dist.init_process_group("gloo", rank=rank, world_size=world)
    run(a.mode, world, rank, a.out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
