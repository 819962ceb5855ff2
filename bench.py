#!/usr/bin/env python3
"""bench.py — bilevel train-step images/sec @512^2 on the pruned SD-2.1 U-Net (BASELINE.json metric), N GPUs of one node.

One "step" = one iteration of the bilevel loop (trainer.py:2769-2816): main step (teacher fwd, student fwd+bwd, three
loss heads, grad all-reduce, AdamW + weight refresh) and, on every 10th iteration, the upper step (teacher cond+uncond
fwd as one 2B batch, student fwd+bwd, negative-guidance loss, all-reduce, upper AdamW).  Synthetic (latent, noise,
timestep, prompt-embed) inputs are resident in HBM before the timed region; weights are random-init SD-2.1 shapes.

Prints ONE JSON line (rank 0).  Extra keys: roofline (dominant kernel, HIP-event timed), cpu_baseline (oracle on host
cores, N=1 only), breakdown.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--batch", type=int, default=8, help="per-GPU batch (BASELINE configs[1]: 8)")
    p.add_argument("--budget", type=float, default=0.55, help="student MACs / teacher MACs (reference 'Pruning Ratio')")
    p.add_argument("--latent", type=int, default=64, help="latent side (64 = 512^2 images)")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--upper_freq", type=int, default=10)
    p.add_argument("--tiny", action="store_true", help="tiny topology (debug only; result is NOT the benchmark)")
    p.add_argument("--no_cpu_baseline", action="store_true")
    p.add_argument("--cpu_baseline_full", action="store_true",
                   help="BASELINE.md 3 protocol: 3 warm-up + 5 timed main steps (median) at all host cores, 1 + 3 at 8 threads, "
                        "plus the upper step (takes ~12 minutes; the default is a bounded sample of the same)")
    p.add_argument("--no_b16", action="store_true", help="skip the extra B=16/GPU measurement (shipped bilevel YAML's batch)")
    p.add_argument("--no_roofline", action="store_true")
    p.add_argument("--no_vae", action="store_true", help="skip the (untimed) VAE-encode extra")
    p.add_argument("--no_graph", action="store_true", help="eager Python launches instead of hipGraph replay")
    return p.parse_args()


def cpu_baseline(budget, latent, tiny, full=False):
    """The oracle's training steps (B=1, fp32, pure-torch CPU restatement of trainer.py:2403-2488 / 2904-3001) on the host
    cores of the GPU box: a reported baseline (`kind: port` - the reference's own trainer cannot run here: no diffusers,
    weights or datasets), not the target.  Protocol = BASELINE.md 3: main step = dense teacher fwd + budget student
    fwd/bwd + 3 loss heads + AdamW; warm-up steps, then the MEDIAN of the timed ones; one upper step (2 teacher fwds +
    student fwd/bwd + upper AdamW) for the bilevel blend 10*B / (10*t_main + t_upper).
    Default (bounded: ~1.5 minutes): a one-step scan over 8 / 16 / 32 / 64 threads, then 1 warm-up + 3 timed main steps + 1
    upper step at the fastest count (`cores` = that count).  --cpu_baseline_full: 3 warm-up + 5 timed there, plus the
    all-cores and 8-thread runs BASELINE.md asks for (profiles/r02_cpu_baseline_full.json keeps one such run)."""
    import platform
    import statistics
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from pdm_ref import weights as oweights, step as ostep
    from pdm_ref.config import UNetConfig as OCfg
    from pdm.models.unet.spec import UNetConfig, arch_vector_for_budget
    ocfg = OCfg.tiny() if tiny else OCfg.sd21()
    cfg = UNetConfig.tiny() if tiny else UNetConfig.sd21()
    T = 13 if tiny else 77
    all_cores = torch.get_num_threads()
    model = platform.processor() or ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    dense = oweights.init_dense_state_dict(ocfg, seed=0)
    av = arch_vector_for_budget(cfg, budget, hw=latent, ctx_len=T)[0]
    psd, info = oweights.prune_state_dict(dense, ocfg, av)
    tinfo = oweights.dense_info(ocfg)
    g = torch.Generator().manual_seed(43)
    lat, noise = torch.randn(1, 4, latent, latent, generator=g), torch.randn(1, 4, latent, latent, generator=g)
    t, ehs = torch.tensor([500]), torch.randn(1, T, ocfg.cross_attention_dim, generator=g)
    empty = torch.randn(1, T, ocfg.cross_attention_dim, generator=g)
    ac = ostep.alphas_cumprod()

    def run(threads, warm, timed):
        torch.set_num_threads(threads)
        P = {k: v.clone().requires_grad_(True) for k, v in psd.items()}
        opt = torch.optim.AdamW(list(P.values()), lr=1e-6, weight_decay=0.0)
        uopt = torch.optim.AdamW(list(P.values()), lr=5e-6, weight_decay=0.0)

        def main_step():
            t0 = time.time()
            ostep.main_step_loss((P, info), (dense, tinfo), ocfg, ac, lat, noise, t, ehs)[0].backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            return time.time() - t0

        def upper_step():
            t0 = time.time()
            ostep.upper_step_loss((P, info), (dense, tinfo), ocfg, ac, lat, noise, t, ehs, empty)[0].backward()
            uopt.step()
            uopt.zero_grad(set_to_none=True)
            return time.time() - t0
        for _ in range(warm):
            main_step()
        ts = [main_step() for _ in range(timed)]
        tm, tu = statistics.median(ts), upper_step()
        return {"threads": threads, "warmup": warm, "timed": timed, "main_step_s_median": round(tm, 3),
                "main_step_s_all": [round(x, 3) for x in ts], "upper_step_s": round(tu, 3),
                "images_per_s_main": round(1.0 / tm, 5), "images_per_s_bilevel": round(10.0 / (10.0 * tm + tu), 5)}
    # torch's CPU kernels do not scale to every hardware thread of a big host (measured on the 128-thread EPYC 9575F of the
    # GPU box: 21.5 s per main step at 128 threads, 5.5 s at 8 - profiles/r02_cpu_baseline_full.json): a short scan picks the
    # thread count the baseline is quoted at, so that it is the CPU's best and not an oversubscription artefact
    scan = {}
    for n in (8, 16, 32, 64):
        if n <= all_cores:
            scan[n] = run(n, 1, 1)["main_step_s_median"]
    best_n = min(scan, key=scan.get) if scan else all_cores
    first = run(best_n, 3 if full else 1, 5 if full else 3)
    out = {"value": first["images_per_s_bilevel"], "unit": "images/s", "cores": best_n, "kind": "port",
           "cpu_model": model, "host_threads": all_cores,
           "sample": f"[bounded sample; BASELINE.md 3's full 3 + 5 protocol at every thread count is kept in "
                     f"profiles/r02_cpu_baseline_full.json (--cpu_baseline_full)] "
                     f"bilevel blend 10/(10 t_main + t_upper) of the pure-torch CPU oracle at B=1, {latent}x{latent} latent, fp32, "
                     f"budget-{budget} student + dense teacher: {first['warmup']} warm-up + {first['timed']} timed main steps "
                     f"(median {first['main_step_s_median']} s) + 1 upper step ({first['upper_step_s']} s) at {best_n} threads = the "
                     f"fastest of a 1-step scan over {sorted(scan)} threads ({all_cores}-thread host)",
           "thread_scan_main_step_s": scan, "best": first}
    if full:      # BASELINE.md 3 also asks for all cores and for 8 threads (the survey container's core count)
        out["all_cores"] = run(all_cores, 3, 5)
        out["threads_8"] = first if best_n == 8 else run(8, 1, 3)
    torch.set_num_threads(all_cores)
    return out


def launch_ranks(a):
    """`python bench.py --gpus N` from a bare shell (no torchrun): start one child process per GPU with the torchrun
    environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), wait for them and return the worst exit code.  The parent
    never initialises the GPU and never execs; rank 0's child prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    rehearsal = os.environ.get("PDMK_BENCH_REHEARSAL") == "1"
    have = torch.cuda.device_count()           # does not initialise the GPU
    if not rehearsal and have < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {have} GPU(s) visible (PDMK_BENCH_REHEARSAL=1 rehearses the "
                         f"N-rank path with every rank on cuda:0 over gloo)")
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs:
            for p_ in list(procs):
                r_ = p_.poll()
                if r_ is None:
                    continue
                procs.remove(p_)
                if r_ != 0:                    # one rank died: the others would hang in a collective
                    rc = rc or r_
                    for q in procs:
                        q.terminate()
            time.sleep(0.2)
    finally:
        for q in procs:
            q.kill()
    return rc


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # rehearsal of the N>1 code path on a 1-GPU box: PDMK_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo
    rehearsal = os.environ.get("PDMK_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    from pdm import _pdmk as k
    from pdm.models.unet.spec import UNetConfig, arch_vector_for_budget, plan_macs
    from pdm.models.unet.unet_2d_conditional import UNet2DConditionModelPruned
    from pdm.training.bilevel import BilevelStepper, GraphedBilevel

    cfg = UNetConfig.tiny() if a.tiny else UNetConfig.sd21()
    T = 13 if a.tiny else 77
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    av, ratio, keep = arch_vector_for_budget(cfg, a.budget, hw=a.latent, ctx_len=T)
    teacher = UNet2DConditionModelPruned(cfg, None, dev, dtype, train=False, seed=0)
    student = UNet2DConditionModelPruned(cfg, av, dev, dtype, train=True, init=False)
    student.load_dense_or_pruned(teacher.state_dict())
    # shipped bilevel config: weights 1.0 / 2.0 / 0.1, snr_gamma 5, upper distillation 1.0, lr 1e-6 / 5e-6
    st = BilevelStepper(student, teacher, w_diff=1.0, w_dist=2.0, w_block=0.1, snr_gamma=5.0, up_w_dist=1.0,
                        lr=1e-6, upper_lr=5e-6, warmup_steps=250, upper_warmup_steps=250)
    Tm = plan_macs(cfg, teacher.blocks, a.latent, T)[0]
    Sm = plan_macs(cfg, student.blocks, a.latent, T)[0]

    B = a.batch
    g = torch.Generator(device=dev).manual_seed(43 + rank)
    nb = 4   # a few distinct resident batches, cycled
    data = [dict(lat=torch.randn(B, 4, a.latent, a.latent, device=dev, generator=g),
                 noise=torch.randn(B, 4, a.latent, a.latent, device=dev, generator=g),
                 t=torch.randint(0, 1000, (B,), device=dev, generator=g),
                 ehs=torch.randn(B, T, cfg.cross_attention_dim, device=dev, generator=g)) for _ in range(nb)]
    empty = torch.randn(1, T, cfg.cross_attention_dim, device=dev, generator=g).expand(B, -1, -1).contiguous()

    def main_iter(i):
        d = data[i % nb]
        st.main_step(d["lat"], d["noise"], d["t"], d["ehs"])
        st.optimizer_step(upper=False)

    def upper_iter(i):
        d = data[(i + 1) % nb]
        st.upper_step(d["lat"], d["noise"], d["t"], d["ehs"], empty)
        st.optimizer_step(upper=True)

    # hipGraph replay (immune to host jitter) on every rank count.  With N > 1 the backward is captured as 12 graphs cut at
    # block boundaries of the tape, and the bucketed RCCL all-reduce of each finished share of the gradient arena is
    # issued on the comm stream between the replays, i.e. it overlaps with the rest of the backward pass.
    # --no_graph: eager launches (all-reduce buckets issued from the backward tape).
    use_graph = not a.no_graph
    graphs = None
    # every rank launches the SAME kernels: rank 0 warms up first (its library times the GEMM candidates of every shape of
    # the step), exports its plan cache, and the other ranks import it before their own warm-up - no rank-to-rank skew from
    # differently tuned plans, and the same split-K sums everywhere
    plan_file = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"pdmk_plans_{os.environ.get('MASTER_PORT', '0')}.txt")

    def warm():
        nonlocal graphs
        if use_graph:              # (--no_graph: plans are made by the untimed warm-up iterations, rank by rank)
            graphs = GraphedBilevel(st, B, 4, a.latent, a.latent, T, cfg.cross_attention_dim,
                                    prefetch=os.environ.get("PDMK_TEACHER_PREFETCH", "1") != "0")
            graphs.capture(bilevel=True)

    if world > 1 and use_graph:
        if rank == 0:
            warm()
            k.plan_export(plan_file)
        dist.barrier()
        if rank != 0:
            k.plan_import(plan_file)
            warm()
        dist.barrier()
        if rank == 0:
            try:
                os.remove(plan_file)
            except OSError:
                pass
    else:
        warm()

    tup = lambda q: (q["lat"], q["noise"], q["t"], q["ehs"])
    prefetch_on = bool(graphs is not None and graphs.prefetch)

    def bilevel_iter(i, base=0):
        """i: iteration of the loop it is called from (upper-step cadence); base + i names the batch (prefetch tokens run on from
        the warm-up loop into the timed one)."""
        d, u = data[(base + i) % nb], data[(base + i + 1) % nb]
        if graphs is None:
            main_iter(i)
            if (i + 1) % a.upper_freq == 0:
                upper_iter(i)
        else:
            # (prefetch mode, GraphedBilevel.prefetch: the next step's batch is announced, as a dataloader one batch ahead would -
            # the upper step's to the main step in front of it, the next main batch to whichever step runs last in this
            # iteration; every timed iteration still holds exactly one main teacher pass, and every tenth one upper pass)
            nxt = dict(next_batch=tup(u), next_id=base + i + 1)
            if (i + 1) % a.upper_freq == 0:
                ub = (u["lat"], u["noise"], u["t"], u["ehs"], empty)
                graphs.main(*tup(d), batch_id=base + i, next_upper=ub, upper_id=("u", base + i))
                graphs.upper(*ub, batch_id=("u", base + i), **nxt)
            else:
                graphs.main(*tup(d), batch_id=base + i, **nxt)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        bilevel_iter(i) if graphs is not None else main_iter(i)
    if a.warmup > 0 and graphs is None:
        upper_iter(0)         # untimed: warm the allocator for the upper step too
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        bilevel_iter(i, a.warmup)
    sync()
    el = time.perf_counter() - t0
    hits_loop = graphs.prefetch_hits if graphs is not None else 0      # teacher passes handed over so far (warm-up + timed loop)
    if world > 1:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())

    # ---- untimed extras on rank 0: per-phase times, roofline of the dominant kernel
    extras = {}
    if rank == 0:
        def timed(fn, n):
            torch.cuda.synchronize()
            s = time.perf_counter()
            for j in range(n):
                fn(j)
            torch.cuda.synchronize()
            return (time.perf_counter() - s) / n
        if world == 1:
            if graphs is not None:
                d0 = data[0]
                # (steady state of the main step alone: every call finds the teacher pass its predecessor queued and queues the next one -
                # one teacher pass per timed call; the first, untimed call primes the chain)
                xmain = lambda j: graphs.main(*tup(data[j % nb]), batch_id=("x", j), next_batch=tup(data[(j + 1) % nb]), next_id=("x", j + 1))
                xmain(-1)
                extras["ms_main_step"] = round(timed(xmain, 4) * 1e3, 2)
                extras["ms_upper_step"] = round(timed(lambda j: graphs.upper(d0["lat"], d0["noise"], d0["t"], d0["ehs"], empty), 2) * 1e3, 2)
                st.defer_reduce = False
            extras["ms_main_step_eager"] = round(timed(main_iter, 3) * 1e3, 2)
            extras["ms_upper_step_eager"] = round(timed(upper_iter, 2) * 1e3, 2)
        extras["launch_mode"] = "eager" if graphs is None else "hipGraph replay"
        extras["teacher_prefetch"] = prefetch_on     # GraphedBilevel: teacher pass of batch t+1 beside step t's backward
        if prefetch_on:      # every main step but the very first, and every upper step, found its teacher pass queued
            n_up = lambda n: sum(1 for i in range(n) if (i + 1) % a.upper_freq == 0)
            extras["teacher_passes_handed_over"] = {"got": hits_loop, "steps": a.warmup + a.steps + n_up(a.warmup) + n_up(a.steps)}
        if graphs is not None:
            extras["graphs_per_main_step"] = len(graphs.g_main.all())
        extras["lockstep_forward"] = bool(st.lockstep)
        if world == 1 and not a.no_vae and not a.tiny:
            # SURVEY 8f N1, NOT part of `value` (SURVEY 8d keeps the VAE off the timed path): what a pixel_values batch adds
            # in front of every step - vae.encode(pixels).latent_dist.sample() * 0.18215 (trainer.py:2405-2406)
            from pdm.models.vae.autoencoder_kl import AutoencoderKL
            vae = AutoencoderKL(None, dev, dtype, seed=0)
            px = torch.rand(B, 3, a.latent * 8, a.latent * 8, device=dev) * 2 - 1
            for _ in range(2):
                vae.encode_latents(px)
            extras["ms_vae_encode_untimed"] = round(timed(lambda j: vae.encode_latents(px), 3) * 1e3, 2)
            del vae, px
            # SURVEY 8f N2, same status: prompt_embeds = text_encoder(input_ids)[0] for the batch's captions (77 tokens)
            from pdm.models.clip.text_encoder import CLIPTextModel
            txt = CLIPTextModel(None, dev, dtype, seed=0)
            tok = torch.randint(0, txt.cfg.vocab_size, (B, 77), device=dev)
            for _ in range(2):
                txt(tok)
            extras["ms_text_encode_untimed"] = round(timed(lambda j: txt(tok), 5) * 1e3, 3)
            del txt, tok
        n_upper = sum(1 for i in range(a.steps) if (i + 1) % a.upper_freq == 0)
        flop_main = 2.0 * (Tm + 3 * Sm) * B
        flop_upper = 2.0 * (2 * Tm + 3 * Sm) * B
        extras["model_tflops_per_gpu"] = round((a.steps * flop_main + n_upper * flop_upper) / el / 1e12, 2)
    roof = None
    if rank == 0 and not a.no_roofline and world == 1:
        # Dominant kernel: one eager main step with every GEMM launch bracketed by HIP events on its launch stream
        # (pdm._pdmk.PROFILE) and labelled with the candidate kernel the library's plan cache picked; the teacher runs
        # in line here (no second stream) so that an event pair times one kernel, not two overlapped ones.
        ts, st.teacher_stream = st.teacher_stream, None
        k.PROFILE = []
        main_iter(0)
        torch.cuda.synchronize()
        prof, k.PROFILE = k.PROFILE, None
        # per-phase times of one eager main / upper step (teacher in line, HIP events at the roctx range boundaries of
        # pdm/utils/roctx.py: no guessing phases from kernel symbol names)
        from pdm.utils import roctx
        for name, fn in (("phases_ms_main_eager_inline_teacher", main_iter), ("phases_ms_upper_eager_inline_teacher", upper_iter)):
            roctx.PHASE_LOG = []
            fn(0)
            torch.cuda.synchronize()
            extras[name] = roctx.summarize(roctx.PHASE_LOG)
            roctx.PHASE_LOG = None
        st.teacher_stream = ts
        if os.environ.get("PDMK_DUMP_GEMM"):
            with open(os.environ["PDMK_DUMP_GEMM"], "w") as f:
                for kind, flops, e0, e1, shp in prof:
                    f.write(json.dumps({"kind": list(kind), "flops": flops, "ms": e0.elapsed_time(e1), "mnk_sk": shp}) + "\n")
        extras["gemm_launches_main_step"] = len(prof)
        gc = student.engine.gn_count          # GroupNorms of the student's last forward pass / with statistics from a GEMM epilogue
        extras["groupnorms_per_forward"], extras["groupnorm_stats_from_gemm_epilogue"] = gc[0], gc[1]
        if student.engine.gn_miss is not None:
            extras["groupnorm_without_epilogue_stats"] = {"student": student.engine.gn_miss, "teacher": teacher.engine.gn_miss}
        extras["gemm_problems_in_grouped_launches"] = sum(kd[4] for kd, *_ in prof if len(kd) > 4 and kd[4] > 1)
        agg, cls = {}, {}
        for kind, flops, e0, e1, shp in prof:
            ms = e0.elapsed_time(e1)
            for d, key in ((agg, kind), (cls, kind[:3])):
                r = d.setdefault(key, [0.0, 0.0, 0])
                r[0] += flops
                r[1] += ms
                r[2] += 1
        class_names = {(0, 0): "linear fwd/dgrad", (1, 0): "conv3x3 fwd/dgrad (implicit GEMM)",
                       (2, 1): "linear wgrad", (2, 2): "conv3x3 wgrad"}
        # candidate 0 = the K-step-32 kernels; rocprofv3's demangler garbles `igemm_kernel<__bf16, a, b, c, 8, 8>`, so the
        # strings it prints for the two weight-gradient instantiations are kept here to look their PMC traffic up
        legacy_sym = {(2, 1): "igemm_kernel<bool _Accum, int, EL, int, E, 0, 8, 8>",
                      (2, 2): "_ZN12_GLOBAL__N_112igemm_kernelIDF16bLi2ELi2ELi0ELi8ELi8EEEv14pdmk_gemm_argsiijj"}
        def sym(kd):
            if kd[3] <= 0:
                return (f"igemm_kernel<__bf16, {kd[1]}, {kd[2]}, 0, 8, 8> [rocprofv3: "
                        f"{legacy_sym.get(kd[1:3], 'igemm_kernel / pdmk_dma::igemm_dma_kernel')}]")
            name = k.candidate_name(kd[1], kd[2], kd[3])
            # a grouped launch (pdmk_gemm_group: several problems in one grid) runs the _group_kernel instantiation
            return name.replace("_kernel<", "_group_kernel<") if (len(kd) > 4 and kd[4] > 1) else name
        dom = max(agg.items(), key=lambda kv: kv[1][1])
        ach = dom[1][0] / (dom[1][1] * 1e-3) / 1e12

        # Two-sided roofline per launch: a GEMM cannot finish before max(MACs / MFMA peak, algorithmic bytes / HBM peak).
        # Algorithmic bytes = each operand once (a conv's input image once, not once per tap) + the output once.  The
        # K = 320 .. 640 Linear layers at 64^2 latents sit BELOW the ridge point (2.5 PFLOP/s / 8 TB/s = 312 FLOP/B): for
        # them the HBM side is the binding one, which the single MFMA fraction above cannot show.
        def two_sided(entries):
            esz = 2 if a.dtype == "bf16" else 4
            pk = (2500e12 if a.dtype == "bf16" else 157.3e12)
            t_m = t_h = t_r = t_meas = t_i = t_3 = 0.0
            n_h = n_i = 0
            b_tot = 0.0
            for kind, flops, e0, e1, shp in entries:
                conv_a, wg = kind[1] == 1, kind[1] == 2
                members = shp if isinstance(shp, list) else [shp]       # a grouped launch lists its problems
                byts = mflop = 0.0
                ti = 0.0
                for (M, N, K, sk, *rd) in members:
                    if wg:      # C[M,N] fp32 += dY[K,M]^T X[K,N]   (conv: X is the image, N = 9 Ci)
                        byts += esz * (K * M + K * (N // 9 if kind[2] == 2 else N)) + 4 * M * N
                    else:       # + the residual / the previous output an accumulating epilogue reads (round 4: it was left out,
                        # which overstated the dominant kernel's traffic_ratio - half of its launches carry a residual)
                        byts += esz * (M * (K // 9 if conv_a else K) + N * K + M * N * (2 if (rd and rd[0]) else 1))
                    mflop += 2.0 * M * N * K
                    # third side: what a CU can take in from L2 into LDS (~70 GB/s per CU, 18 TB/s chip-wide: MI355X_MICROARCH.md
                    # "Indexed rows: gather into LDS", tools/small_gemm_sweep.py).  An output tile BM x BN needs (BM + BN) K
                    # operand elements whatever the kernel (a halo-staged 3x3 conv reads its activation patch once per 9 taps);
                    # best case over the tile shapes a 512-thread workgroup can hold, with tiles spread over 256 CUs
                    tim = float("inf")
                    for bm in (64, 128, 256):
                        for bn in (64, 128, 160, 256):
                            tiles = -(-M // bm) * -(-N // bn)
                            a_el = bm * (K / 9.0 if (conv_a or (wg and kind[2] == 2)) else K)
                            per_cu = -(-tiles // 256) * (a_el + bn * K) * esz
                            tim = min(tim, per_cu / 70e9)
                    ti += tim
                tm, th = mflop / pk, byts / 8e12
                b_tot += byts
                t_m, t_h, t_i = t_m + tm, t_h + th, t_i + ti
                t_r, t_3 = t_r + max(tm, th), t_3 + max(tm, th, ti)
                n_h += th > tm
                n_i += ti > max(tm, th)
                t_meas += e0.elapsed_time(e1) * 1e-3
            return {"mfma_floor_ms": round(t_m * 1e3, 3), "hbm_floor_ms": round(t_h * 1e3, 3),
                    "roofline_floor_ms": round(t_r * 1e3, 3), "measured_ms": round(t_meas * 1e3, 3),
                    "frac": round(t_r / t_meas, 4), "launches": len(entries), "hbm_bound_launches": int(n_h),
                    "algorithmic_bytes_per_launch": round(b_tot / max(len(entries), 1)),
                    "l2_intake_floor_ms": round(t_i * 1e3, 3), "three_sided_floor_ms": round(t_3 * 1e3, 3),
                    "three_sided_frac": round(t_3 / t_meas, 4), "intake_bound_launches": int(n_i)}
        peak = 2500.0 if a.dtype == "bf16" else 157.3
        traffic = None
        import glob
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
        tfile = tfiles[-1] if tfiles else ""
        if tfile and not a.tiny:
            # HBM bytes per launch of this kernel symbol from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
            # this same command (tools/summarize_pmc.py; gfx950 2x FETCH_SIZE correction applied); counters cannot be read live
            rec = json.load(open(tfile)).get(sym(dom[0]) if dom[0][3] > 0 else legacy_sym.get(dom[0][1:3], ""))
            traffic = rec and round(rec["hbm_bytes_per_launch"])
        top = sorted(agg.items(), key=lambda kv: -kv[1][1])[:6]
        dom_ts = two_sided([p_ for p_ in prof if p_[0] == dom[0]])
        alg_b = dom_ts["algorithmic_bytes_per_launch"]
        roof = {"bound": "mfma", "kernel": sym(dom[0]), "kernel_class": class_names[dom[0][1:3]],
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "traffic": traffic, "traffic_unit": "HBM bytes/launch",
                # algorithmic bytes: every operand and the output once (the same formula as two_sided.hbm_floor_ms)
                "algorithmic_bytes": alg_b, "traffic_ratio": round(traffic / alg_b, 3) if (traffic and alg_b) else None,
                "traffic_source": (f"committed profile profiles/{os.path.basename(tfile)} (separate rocprofv3 --pmc FETCH_SIZE / "
                                   f"WRITE_SIZE passes of this command, gfx950 2x FETCH_SIZE correction, keyed by kernel symbol; "
                                   f"not measured in this run: counters cannot be read live)") if traffic else None,
                # what the ratio is made of for the halo conv (isolated probes, profiles/r04_traffic_probe_conv.txt): the (R + 2) / R halo
                # rows of an R-image-row tile fetched by both neighbours at once and one copy of the weights per XCD L2; FETCH_SIZE also
                # counts requests the Infinity Cache serves, so this is fabric traffic - an upper bound of HBM traffic
                "traffic_note": ("halo rows (x (R+2)/R per tile of R image rows) + one weight copy per XCD L2; fabric requests incl. "
                                 "Infinity-Cache hits (DESIGN.md 5.0)") if (traffic and "conv_halo" in sym(dom[0])) else None,
                "launches": dom[1][2], "avg_launch_ms": round(dom[1][1] / dom[1][2], 4),
                "two_sided": {"note": "per launch max(MACs/2.5 PFLOP/s, algorithmic bytes/8 TB/s) summed, over measured time; "
                                      "three_sided adds the L2->LDS operand intake of the best 512-thread tile at 70 GB/s per CU",
                              "dominant_kernel": dom_ts,
                              "all_gemms": two_sided(prof)},
                "avg_launch_gflop": round(dom[1][0] / dom[1][2] / 1e9, 3),
                "gemm_classes": {class_names[kd[1:3]]: {"tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 2), "ms": round(v[1], 2),
                                                        "launches": v[2]} for kd, v in cls.items()},
                "top_kernels": {sym(kd) + " | " + class_names[kd[1:3]]: {"tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 2),
                                                                        "ms": round(v[1], 2), "launches": v[2]}
                                for kd, v in top}}
    if rank == 0 and world == 1 and not a.no_b16 and not a.tiny and B != 16 and graphs is not None:
        # the shipped bilevel YAML's per-GPU batch (configs/baselines/sd-2-1_coco_aptp_both_512_bilevel.yaml:48), same cadence,
        # same protocol, NOT the bench value (configs[1] is quoted at B = 8)
        graphs.close()
        graphs = None
        torch.cuda.empty_cache()
        B2 = 16
        d2 = [dict(lat=torch.randn(B2, 4, a.latent, a.latent, device=dev, generator=g),
                   noise=torch.randn(B2, 4, a.latent, a.latent, device=dev, generator=g),
                   t=torch.randint(0, 1000, (B2,), device=dev, generator=g),
                   ehs=torch.randn(B2, T, cfg.cross_attention_dim, device=dev, generator=g)) for _ in range(2)]
        e2 = empty[:1].expand(B2, -1, -1).contiguous()
        g2 = GraphedBilevel(st, B2, 4, a.latent, a.latent, T, cfg.cross_attention_dim,
                            prefetch=os.environ.get("PDMK_TEACHER_PREFETCH", "1") != "0")
        g2.capture(bilevel=True)

        def it2(i, base=0):          # the protocol of bilevel_iter above
            d, u = d2[(base + i) % 2], d2[(base + i + 1) % 2]
            nxt = dict(next_batch=tup(u), next_id=base + i + 1)
            if (i + 1) % a.upper_freq == 0:
                ub = (d["lat"], d["noise"], d["t"], d["ehs"], e2)
                g2.main(*tup(d), batch_id=base + i, next_upper=ub, upper_id=("u", base + i))
                g2.upper(*ub, batch_id=("u", base + i), **nxt)
            else:
                g2.main(*tup(d), batch_id=base + i, **nxt)
        for i in range(2):
            it2(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            it2(i, 2)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t0
        extras["b16"] = {"images_per_s": round(10 * B2 / el2, 2), "ms_per_step": round(el2 / 10 * 1e3, 2), "batch": B2,
                         "steps": 10, "warmup": 2}
        del g2, d2
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        del data
        torch.cuda.empty_cache()
        cpu = cpu_baseline(a.budget, a.latent, a.tiny, full=a.cpu_baseline_full)

    if rank == 0:
        value = a.steps * B * world / el
        out = {"metric": "bilevel train-step images/sec @512^2 pruned-SD2.1", "value": round(value, 3),
               "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(el / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": f"BASELINE configs[1] model+batch (MAC-budget-{a.budget} SD-2.1 student [ratio "
                                      f"{ratio:.3f}, {Sm / 1e9:.1f} GMAC/img] + dense teacher [{Tm / 1e9:.1f} GMAC/img], "
                                      f"B={B}/GPU, {a.latent}x{a.latent} latents = {a.latent * 8}^2 px, losses 1.0 ddpm(min-SNR 5) "
                                      f"+ 2.0 distill + 0.1 block) under the bilevel cadence of configs[2] (upper "
                                      f"concept-suppression step every {a.upper_freq}th iteration, second AdamW)"
                                      + (" [TINY DEBUG TOPOLOGY - not the benchmark]" if a.tiny else ""),
                          "global_batch": B * world, "parallelism": f"dp{world}", "student_params": student.num_parameters(),
                          "weights": "random-init",
                          # "activation dtype" unless PDMK_ATTN_FP8=1 rounds Q/K/V to e4m3 (configs[4]; never the default line)
                          "attention_precision": student.attention_precision or "activation dtype",
                          # what the ranks really talked over (the driver's SCALE record can be checked against it)
                          "dist_backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if world > 1 else "none",
                          "rccl_ranks": (dist.get_world_size() if (world > 1 and dist.get_backend() == "nccl") else 0),
                          "dp_mode": st.reducer.mode + ("/native-comm" if st.reducer.comm is not None else ""),
                          # one batch of look-ahead: the frozen teacher's pass over batch t+1 runs beside step t's backward; every
                          # timed iteration holds exactly one teacher pass (Trainer: training.teacher_prefetch)
                          "teacher_prefetch": prefetch_on,
                          "curve_note": "the shipped bilevel YAML runs B=16/GPU: the N=1 point of THAT curve is extras.b16 "
                                        "(N>1: rerun with --batch 16); `value` is configs[1]'s B=8/GPU"},
               "roofline": roof, "cpu_baseline": cpu, "extras": extras}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
