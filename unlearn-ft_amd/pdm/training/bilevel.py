"""The bilevel fine-tune / unlearn step on the MI355X engine: forward diffusion, teacher + student passes, the four loss
heads, hand-written backward, data-parallel gradient reduction and the two fused AdamW optimisers.

Reference arithmetic:
  main step    pdm/training/trainer.py:2403-2488  (DDPM min-SNR(gamma) + w_block * block-feature MSE + w_dist * output MSE)
  upper step   trainer.py:2904-3001               (ESD-style negative guidance target e_u - (e_c - e_u))
  loop cadence trainer.py:2769-2816               (upper step every `upper_step_freq` iterations, its own AdamW + LR)
  optimisers   trainer.py:265-284, 2695-2717      (torch.optim.AdamW semantics), LR trainer.py:436-443, 2666-2674
  DDP          trainer.py:117-129, 2782, 2808     (gradient mean over ranks)
"""
import gc
import math
import os

import torch
import torch.distributed as dist

from .. import _pdmk as k
from ..models.unet.spec import padc
from ..utils.metric_utils import alphas_cumprod_sd, min_snr_weight_table
from ..utils.roctx import phase

BLOCK_KEYS = ("d0", "d1", "d2", "d3", "m", "u0", "u1", "u2", "u3")


class FusedAdamW:
    """torch.optim.AdamW over the flat fp32 arena, one kernel launch (pdmk_adamw)."""

    def __init__(self, store, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, warmup_steps=0, sched_mult=1):
        dev = store.master.device
        self.store = store
        self.base_lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.m = torch.zeros_like(store.master)
        self.v = torch.zeros_like(store.master)
        # [lr, 1 - b1^t, 1 - b2^t] live in ONE device vector refreshed per step by an asynchronous copy out of a ring of
        # pinned host rows (a pageable H2D copy would make the host wait for the stream to drain: no run-ahead, the GPU
        # idles between iterations); lr_dev / bc_dev are views of it
        self._sched_dev = torch.tensor([0.0, 1.0, 1.0], device=dev)    # bc = 1, not 0: a launch() before the first prepare()
        self.lr_dev, self.bc_dev = self._sched_dev[:1], self._sched_dev[1:]      # (graph warm-up) must not divide by 0
        self._sched_host = torch.zeros(32, 3).pin_memory() if dev.type == "cuda" else torch.zeros(32, 3)
        # one event per ring row, recorded behind that row's copy: a row is rewritten only after its copy has executed (a
        # host that replays graphs can run more than 32 steps ahead of the device; in the steady state the wait is free)
        self._sched_ev = [None] * self._sched_host.shape[0]
        self.t = 0                     # optimiser steps taken
        self.sched_k = 0               # scheduler.step() calls (accelerate steps it `sched_mult`=W times per opt step)
        self.warmup = warmup_steps     # already multiplied by W by the caller (trainer.py:436-443)
        self.sched_mult = sched_mult

    def current_lr(self):
        if self.warmup > 0:
            return self.base_lr * min(1.0, self.sched_k / float(self.warmup))
        return self.base_lr

    def prepare(self):
        """Host-side part of a step (never captured in a graph): advance t, publish lr and bias corrections."""
        self.t += 1
        lr = self.current_lr()
        slot = self.t % self._sched_host.shape[0]
        row = self._sched_host[slot]
        if self._sched_ev[slot] is not None:
            self._sched_ev[slot].synchronize()      # the copy that last read this pinned row has executed
        row[0], row[1], row[2] = lr, 1 - self.betas[0] ** self.t, 1 - self.betas[1] ** self.t
        self._sched_dev.copy_(row, non_blocking=True)
        if self._sched_dev.is_cuda:
            if self._sched_ev[slot] is None:
                self._sched_ev[slot] = torch.cuda.Event()
            self._sched_ev[slot].record()
        self.sched_k += self.sched_mult
        return lr

    def launch(self, grad_scale=1.0, zero_grad=True):
        """Device-side part (graph-capturable): fused AdamW + refresh of the compute copies."""
        s = self.store
        fused = s.dtype == torch.bfloat16
        with phase("adamw"):
            k.adamw(s.master, s.grad, self.m, self.v, s.total, self.lr_dev, self.betas[0], self.betas[1], self.eps,
                    self.wd, self.bc_dev, grad_scale, zero_grad, w_bf16=s.w if fused else None)
            s.refresh(w_is_fresh=fused, wt=not s.defer_wt)

    def launch_range(self, lo, hi, grad_scale=1.0, zero_grad=True):
        """AdamW on arena slice [lo, hi) only (no refresh of the transposed copies): lets the update of layers whose
        gradients are final stream under the rest of the backward pass (HBM-bound kernel beside MFMA-bound ones)."""
        if hi <= lo:
            return
        s = self.store
        fused = s.dtype == torch.bfloat16
        k.adamw(s.master[lo:hi], s.grad[lo:hi], self.m[lo:hi], self.v[lo:hi], hi - lo, self.lr_dev, self.betas[0],
                self.betas[1], self.eps, self.wd, self.bc_dev, grad_scale, zero_grad, w_bf16=s.w[lo:hi] if fused else None)

    def step(self, grad_scale=1.0, zero_grad=True):
        lr = self.prepare()
        self.launch(grad_scale, zero_grad)
        return lr

    # ---- checkpoint interchange: the files accelerator.save_state writes for a torch AdamW + LambdaLR pair
    # (pdm/training/trainer.py:452-477: optimizer.bin / optimizer_1.bin, scheduler.bin / scheduler_1.bin)
    def state_dict(self):
        """`torch.optim.AdamW.state_dict()` layout: per-parameter {step, exp_avg, exp_avg_sq} in the reference's shapes
        (diffusers names, pruned), keyed by the parameter's index in the reference module's `.parameters()` order."""
        from ..models.unet.params import reference_param_order
        s = self.store
        order = reference_param_order(list(s.state_dict_names()))
        state = {}
        if self.t > 0:
            m, v = s.state_dict(arena=self.m), s.state_dict(arena=self.v)
            step = torch.tensor(float(self.t))
            state = {i: {"step": step.clone(), "exp_avg": m[n], "exp_avg_sq": v[n]} for i, n in enumerate(order)}
        group = {"lr": self.current_lr(), "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd,
                 "amsgrad": False, "foreach": None, "maximize": False, "capturable": False, "differentiable": False,
                 "fused": None, "initial_lr": self.base_lr, "params": list(range(len(order)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        from ..models.unet.params import reference_param_order
        s = self.store
        order = reference_param_order(list(s.state_dict_names()))
        st = sd["state"]
        if len(st) == 0:
            self.m.zero_()
            self.v.zero_()
            self.t = 0
            return
        if len(st) != len(order):
            raise ValueError(f"optimizer state has {len(st)} parameters, this model has {len(order)}")
        s.load_state_dict({n: st[i]["exp_avg"] for i, n in enumerate(order)}, arena=self.m)
        s.load_state_dict({n: st[i]["exp_avg_sq"] for i, n in enumerate(order)}, arena=self.v)
        self.t = int(round(float(st[0]["step"])))

    def scheduler_state_dict(self):
        """`LambdaLR.state_dict()` layout of diffusers' constant_with_warmup scheduler (trainer.py:436-443)."""
        lr = self.current_lr()
        return {"base_lrs": [self.base_lr], "last_epoch": self.sched_k, "verbose": False, "_step_count": self.sched_k + 1,
                "_get_lr_called_within_step": False, "_last_lr": [lr], "lr_lambdas": [None]}

    def load_scheduler_state_dict(self, sd):
        self.sched_k = int(sd["last_epoch"])


class GradReducer:
    """Data-parallel mean of the flat gradient arena over RCCL (`nccl` backend) / gloo.

    Reference = DDP's bucketed all-reduce inside accelerator.backward (trainer.py:2782, 2808).  Here the arena is
    reduced in `bucket_mb` slices issued on the dedicated comm stream as soon as the backward pass has produced them (the
    arena is laid out in forward order, so the tail is final first); the division by world size is folded into AdamW.

    mode "allreduce" (default): one all-reduce per bucket.  mode "rs_ag" (PDMK_DP_MODE=rs_ag; EXPERIMENTAL -
    its RCCL branch has never run on more than one GPU): every bucket as reduce-scatter + all-gather of world
    equal shares - over RCCL each rank receives its share directly from its xGMI peers; a bucket's last (n mod world) elements
    ride in a small all-reduce.  Both modes give the same sums.
    transport: torch.distributed's process group (default) or the library's own communicator (PDMK_COMM=native).
    The torch.distributed branch of rs_ag receives this rank's share in a SCRATCH buffer (`_share`) and gathers out of it:
    whether `reduce_scatter_tensor(out, in)` / `all_gather_into_tensor(out, in)` accept an output that aliases the input
    depends on the backend (NCCL / RCCL document exactly the in-place placement `in + rank * count`; others do not), and
    1 / world of a bucket is cheap.  The native communicator uses RCCL's documented in-place form (comm.hip)."""

    def __init__(self, store, bucket_mb=64, mode=None):
        self.store = store
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.bucket = bucket_mb * (1 << 20) // 4
        self.mode = mode or os.environ.get("PDMK_DP_MODE", "allreduce")
        if self.mode not in ("allreduce", "rs_ag"):
            raise ValueError(f"PDMK_DP_MODE={self.mode!r}: expected 'allreduce' or 'rs_ag'")
        self.stream = k.role_stream(store.master.device, "comm") if (self.world > 1 and store.master.is_cuda) else None
        self.next_hi = store.total
        self.flush_cb = None               # set by the stepper: the engine's flush_pending()
        self.n_collectives = 0             # collectives issued since begin() (tests / bench bookkeeping)
        self.backend = dist.get_backend() if self.world > 1 else "none"
        # PDMK_COMM=native: the collectives go through the library's own communicator handle (pdmk_comm_t, RCCL bound inside
        # libpdmk.so) instead of torch.distributed's process group; the 128-byte id travels over torch.distributed once.
        self.comm = None
        if self.world > 1 and store.master.is_cuda and os.environ.get("PDMK_COMM") == "native":
            box = [k.Comm.unique_id() if dist.get_rank() == 0 else None]
            dist.broadcast_object_list(box, src=0)
            self.comm = k.Comm(box[0], dist.get_rank(), self.world)
            # the handle and torch.distributed must describe the same job: share r of a bucket belongs to rank r of BOTH
            if (self.comm.world_size(), self.comm.rank_id()) != (self.world, self.rank):
                raise RuntimeError(f"pdmk_comm_t is rank {self.comm.rank_id()} of {self.comm.world_size()}, "
                                   f"torch.distributed says rank {self.rank} of {self.world}")
        self._share = None                 # rs_ag over torch.distributed: this rank's share of a bucket (scratch)

    def begin(self):
        self.next_hi = self.store.total
        self.n_collectives = 0

    def ready_down_to(self, lo):
        """Everything in [lo, total) is final: launch whole buckets from the tail (called from the backward tape)."""
        if self.world == 1:
            return
        lo = max(lo, 0)
        while self.next_hi - lo >= self.bucket:
            if self.flush_cb is not None:
                self.flush_cb()            # deferred norm-affine gradient reductions of the blocks behind `lo`
            self._launch(self.next_hi - self.bucket, self.next_hi)
            self.next_hi -= self.bucket

    # ---- one bucket
    def _all_reduce(self, t):
        self.n_collectives += 1
        if self.comm is not None:
            self.comm.all_reduce_sum_(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def _rs_ag(self, t):
        """Sum of bucket `t` over the ranks as reduce-scatter + all-gather of `world` equal shares (+ a tail all-reduce)."""
        W = self.world
        n = t.numel() - t.numel() % W
        if n:
            body = t[:n]
            per = n // W
            self.n_collectives += 2
            if self.comm is not None:
                self.comm.reduce_scatter_sum_(body)
                self.comm.all_gather_(body)
            elif self.backend == "gloo":        # gloo has no reduce-scatter: the same data movement share by share
                for r in range(W):
                    dist.reduce(body[r * per:(r + 1) * per], dst=r, op=dist.ReduceOp.SUM)
                for r in range(W):
                    dist.broadcast(body[r * per:(r + 1) * per], src=r)
            else:
                if self._share is None or self._share.numel() < per or self._share.device != t.device:
                    self._share = torch.empty(max(per, (self.bucket + W - 1) // W), dtype=t.dtype, device=t.device)
                share = self._share[:per]              # never aliases the bucket (see the class docstring)
                dist.reduce_scatter_tensor(share, body, op=dist.ReduceOp.SUM)
                dist.all_gather_into_tensor(body, share)
        if n < t.numel():
            self._all_reduce(t[n:])

    def _launch(self, lo, hi):
        g = self.store.grad[lo:hi]
        reduce_ = self._rs_ag if self.mode == "rs_ag" else self._all_reduce
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream), phase("allreduce"):
                reduce_(g)
        else:
            with phase("allreduce"):
                reduce_(g)

    def finish(self):
        if self.world == 1:
            return 1.0
        if self.next_hi > 0:
            if self.flush_cb is not None:
                self.flush_cb()
            self._launch(0, self.next_hi)
            self.next_hi = 0
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        return 1.0 / self.world


class _StepCtx:
    """What the pieces of one step hand to each other (forward inputs / outputs of the student)."""
    __slots__ = ("B", "C", "H", "W", "timesteps", "noisy", "target", "ehs", "pred", "acts")


class BilevelStepper:
    """The step as capturable pieces.  Eager mode (`main_step` / `upper_step`) composes them over the dedicated role
    streams (teacher pass and dgrad-copy refresh beside the student forward); graph mode (GraphedBilevel) captures every
    piece as a SINGLE-STREAM hipGraph and does the same composition between the graphs - no piece switches streams when
    `in_graph` is set."""

    def __init__(self, student, teacher, *, w_diff=1.0, w_dist=2.0, w_block=0.1, snr_gamma=5.0, up_w_diff=0.0,
                 up_w_dist=1.0, up_w_block=0.0, prediction_type="v_prediction", lr=1e-6, upper_lr=5e-6,
                 betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, warmup_steps=0, upper_warmup_steps=0,
                 bilevel=True, bucket_mb=64):
        if prediction_type not in ("v_prediction", "epsilon"):
            raise ValueError(f"Unknown prediction type {prediction_type}")        # trainer.py:2445
        self.student, self.teacher = student, teacher
        self.dev = student.device
        self.w = dict(diff=w_diff, dist=w_dist, block=w_block, up_diff=up_w_diff, up_dist=up_w_dist, up_block=up_w_block)
        self.prediction_type = prediction_type
        ac = alphas_cumprod_sd()
        self.sqrt_acp = ac.sqrt().contiguous().to(self.dev)
        self.sqrt_1macp = (1.0 - ac).sqrt().contiguous().to(self.dev)
        self.snr_w = (min_snr_weight_table(ac, snr_gamma, prediction_type == "v_prediction") if snr_gamma is not None
                      else torch.ones(1000)).to(self.dev)
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.world = world
        self.opt = FusedAdamW(student.store, lr, betas, eps, weight_decay, warmup_steps * world, world)
        self.upper_opt = FusedAdamW(student.store, upper_lr, betas, eps, weight_decay, upper_warmup_steps * world,
                                    world) if bilevel else None
        self.reducer = GradReducer(student.store, bucket_mb)
        self.reducer.flush_cb = student.engine.flush_pending
        self.defer_reduce = False
        self.accum = 1                 # training.gradient_accumulation_steps: accelerate's backward divides every loss by it (Trainer.train)
        self.segment_cb = None
        self.in_graph = False          # GraphedBilevel: the pieces run under stream capture, on ONE stream
        # lockstep forward (PDMK_LOCKSTEP=1; OFF by default): the frozen teacher and the student run the same layer sequence
        # on independent data (trainer.py:2446-2459, 2951-2954); both forwards are recorded and issued side by side on ONE
        # stream, layer l of both as one grouped launch where the library has a kernel for the pair (pdmk_gemm_group).
        # Measured on one MI355X at B = 8 (same-box A/B): 45.8 ms per main step against 42.9 ms with the two
        # streams - the grouped GEMM launches (121 of them) win less than the two-stream overlap of everything else loses.
        self.lockstep = os.environ.get("PDMK_LOCKSTEP", "0") == "1" and student.dtype == torch.bfloat16
        self._gscale = 1.0 / world
        # the frozen teacher pass and the student forward are independent until the loss heads: two HIP streams, so that the
        # small-grid layers of one fill the CUs the other leaves idle (PDMK_TEACHER_STREAM=0 runs the teacher in line).
        # Every role has ONE dedicated stream per process (k.role_stream): never a pooled torch stream, which would alias
        # another role after a few stepper instances.
        cuda = self.dev.type == "cuda"
        t_hi = os.environ.get("PDMK_TEACHER_PRIO", "0") == "1"     # (A/B knob: the role streams are low-priority streams by default)
        self.teacher_stream = (k.role_stream(self.dev, "teacher_hi" if t_hi else "teacher", high_priority=t_hi)
                               if cuda and os.environ.get("PDMK_TEACHER_STREAM", "1") != "0" else None)
        self.losses = torch.zeros(4, device=self.dev, dtype=torch.float64)   # diff, dist, block, (unused); zeroed by k.zero_
        # transposed (dgrad) weight copies are refreshed at the START of the next training step, beside its forward, instead
        # of at the end of the optimiser step (PDMK_DEFER_WT=0: refresh with the optimiser as before)
        self.wt_stream = k.role_stream(self.dev, "wt") if cuda else None
        self._wt_pending = False
        student.store.defer_wt = os.environ.get("PDMK_DEFER_WT", "1") != "0"

    # ------------------------------------------------------------------ pieces
    def _diffuse(self, latents, noise, timesteps, want_target):
        B, C, H, W = latents.shape
        cp = padc(C)
        dt = self.student.dtype
        noisy = torch.empty((B * H * W, cp), device=self.dev, dtype=dt)
        target = torch.empty((B * H * W, cp), device=self.dev, dtype=torch.float32) if want_target else None
        k.add_noise_velocity(latents, noise, timesteps, self.sqrt_acp, self.sqrt_1macp, noisy, target, B, C, H * W, cp)
        if want_target and self.prediction_type == "epsilon":
            k.nchw_to_nhwc(noise, target, B, C, H * W, cp)
        return noisy, target

    def _ehs2d(self, e):
        """[B, T, D] prompt embeddings (fp32 in the reference's batch schema, data_utils.py:286-312) -> [B * T, D] rows in the
        compute dtype; the cast is pdmk_cast_permute, so a replayed step holds no torch arithmetic kernel for it."""
        e = e.to(self.dev)
        rows, cols = e.shape[0] * e.shape[1], e.shape[2]
        if e.dtype == torch.float32 and self.student.dtype != torch.float32 and e.is_contiguous():
            out = torch.empty((rows, cols), device=self.dev, dtype=self.student.dtype)
            k.cast_permute(e, out, rows * cols, 1, 1, 0)
            return out
        return e.to(self.student.dtype).reshape(rows, cols).contiguous()

    def _block_loss(self, acts_s, acts_t, B, weight, uncond_half=False, seed=True):
        """(1/9) sum_k mse(student_k, teacher_k) (trainer.py:2475-2481) and its gradient seeds; uncond_half: the upper step
        reads the UNCONDITIONAL half (rows [M, 2M) of every activation) of the 2B teacher batch - the reference's teacher
        hooks hold the last teacher call, trainer.py:2951-2954."""
        # when every layer of a block is dropped and it has no sampler, two hook keys hold the SAME activation (e.g. both
        # resnets of down_blocks.3 dropped: acts['d3'] is acts['d2']): its gradient seed is the sum of both terms
        seeded = set()
        for key in BLOCK_KEYS:
            a, b = acts_s[key], acts_t[key]
            M, C = a.t.shape
            bt = b.t[M:2 * M] if uncond_half else b.t[:M]
            n = len(BLOCK_KEYS) * M * C
            again = id(a) in seeded
            if seed and weight > 0 and not again:
                a.g = torch.empty_like(a.t)
                seeded.add(id(a))
            k.mse_fwd_bwd(a.t, bt, None, self.losses, 2, a.g if (seed and weight > 0) else None, B, M // B, C,
                          a.t.stride(0), bt.stride(0), C, 1.0 / n, 2.0 * weight / n, again)

    def begin_wt_refresh(self):
        """The dgrad copies `wt` (W^T, flipped conv taps) of the weights the last optimiser step wrote are only read by the
        backward pass: their refresh (one 3.4 GB HBM-bound pass) runs on the wt stream beside the forward.  Never captured:
        GraphedBilevel calls it between its graphs."""
        store = self.student.store
        if not store.defer_wt or self.wt_stream is None:
            return
        self.wt_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.wt_stream), phase("wt_refresh"):
            store.refresh_wt()
        self._wt_pending = True

    def join_wt_refresh(self):
        if self._wt_pending:
            torch.cuda.current_stream().wait_stream(self.wt_stream)
            self._wt_pending = False

    def backward(self):
        """Backward pass of the student + gradient reduction; returns the gradient scale for AdamW (1 / world)."""
        if not self.in_graph:
            self.join_wt_refresh()
        if self.defer_reduce:          # graph mode: the all-reduce is issued by the caller between captured graphs
            self.student.engine.grad_ready_cb = self.segment_cb      # None, or GraphedBilevel's capture-segment switch
            self.student.engine.backward()
            return 1.0 / (self.world * self.accum)
        self.reducer.begin()
        self.student.engine.grad_ready_cb = self.reducer.ready_down_to
        self.student.engine.backward()
        return self.reducer.finish() / self.accum

    def reduce_now(self):
        self.reducer.begin()
        return self.reducer.finish()

    def teacher_pass(self, latents, noise, timesteps, prompt_embeds, input_noise=None):
        """The frozen teacher's part of a main step on its own: forward diffusion + dense forward (trainer.py:2446-2448).
        Returns (pred Act, {block key: Act}).  It reads nothing the student writes."""
        B, C, H, W = latents.shape
        with phase("fwd_teacher"):
            noisy, _ = self._diffuse(latents, noise if input_noise is None else input_noise, timesteps, False)
            return self.teacher.forward_nhwc(noisy, timesteps, self._ehs2d(prompt_embeds), B, H, W, train=False)

    def upper_teacher_pass(self, latents, noise, timesteps, prompt_embeds, empty_prompt_embeds):
        """Teacher cond + uncond predictions of the upper step as ONE batch of 2B (trainer.py:2951-2954)."""
        B, C, H, W = latents.shape
        with phase("fwd_teacher"):
            # (the doubled batch is formed from the INPUTS: nothing here may read the output of a kernel of this pass on the
            # host side of the stream - under lockstep recording the kernels are issued later)
            t2 = torch.cat([timesteps, timesteps], 0)
            noisy2, _ = self._diffuse(torch.cat([latents, latents], 0), torch.cat([noise, noise], 0), t2, False)
            # (... which is why the two prompt batches are concatenated BEFORE the cast: _ehs2d's cast is a recorded kernel, and a
            # torch.cat of its two outputs would run at once, on memory the cast has not written yet)
            ehs2 = self._ehs2d(torch.cat([prompt_embeds.to(self.dev), empty_prompt_embeds.to(self.dev)], 0))
            return self.teacher.forward_nhwc(noisy2, t2, ehs2, 2 * B, H, W, train=False)

    def student_forward(self, latents, noise, timesteps, prompt_embeds, train=True, input_noise=None, want_target=True):
        """Forward diffusion (trainer.py:2409-2445) + student forward; zeroes the loss accumulators."""
        ctx = _StepCtx()
        ctx.B, ctx.C, ctx.H, ctx.W = latents.shape
        ctx.timesteps = timesteps
        with phase("diffuse"):
            ctx.noisy, ctx.target = self._diffuse(latents, noise, timesteps, want_target)
            if input_noise is not None:
                ctx.noisy, _ = self._diffuse(latents, input_noise, timesteps, False)
            ctx.ehs = self._ehs2d(prompt_embeds)
            k.zero_(self.losses)
        with phase("fwd_student"):
            ctx.pred, ctx.acts = self.student.forward_nhwc(ctx.noisy, timesteps, ctx.ehs, ctx.B, ctx.H, ctx.W, train=train)
        self.last_pred = ctx.pred
        return ctx

    def main_loss_heads(self, ctx, teacher_out, backward=True):
        """DDPM min-SNR head + output distillation + block-feature distillation (trainer.py:2451-2488) and their seeds."""
        w = self.w
        B, C, H, W = ctx.B, ctx.C, ctx.H, ctx.W
        pred, target = ctx.pred, ctx.target
        with phase("loss"):
            wb = self.snr_w[ctx.timesteps].contiguous()
            HW, cp, n = H * W, pred.t.shape[1], B * H * W * C
            k.mse_fwd(pred.t, target, wb, self.losses, 0, B, HW, C, cp, cp, 1.0 / n)
            if w["dist"] > 0:
                k.mse_fwd(pred.t, teacher_out[0].t, None, self.losses, 1, B, HW, C, cp, cp, 1.0 / n)
            if backward:
                pred.g = k.zeros(tuple(pred.t.shape), pred.t.device, pred.t.dtype)
                k.mse_bwd(pred.t, target, wb, pred.g, B, HW, C, cp, cp, cp, 2.0 * w["diff"] / n, False)
                if w["dist"] > 0:
                    k.mse_bwd(pred.t, teacher_out[0].t, None, pred.g, B, HW, C, cp, cp, cp, 2.0 * w["dist"] / n, True)
            if w["block"] > 0:
                self._block_loss(ctx.acts, teacher_out[1], B, w["block"], seed=backward)

    def upper_loss_heads(self, ctx, teacher_out, backward=True):
        """mse(student(x_t, c), e_u - (e_c - e_u)) (trainer.py:2983-3001) and its seed."""
        w = self.w
        B, C, H, W = ctx.B, ctx.C, ctx.H, ctx.W
        pred, (pred_t, acts_t) = ctx.pred, teacher_out
        M = B * H * W
        e_c, e_u = pred_t.t[:M], pred_t.t[M:]
        k.axpby(e_c, e_u, -1.0, 2.0)               # e_u <- 2 e_u - e_c  == e_u - (e_c - e_u)
        cp, n = pred.t.shape[1], M * C
        k.mse_fwd(pred.t, e_u, None, self.losses, 1, B, H * W, C, cp, cp, 1.0 / n)
        if backward:
            pred.g = k.zeros(tuple(pred.t.shape), pred.t.device, pred.t.dtype)
            k.mse_bwd(pred.t, e_u, None, pred.g, B, H * W, C, cp, cp, cp, 2.0 * w["up_dist"] / n, False)
        if w["up_block"] > 0:
            self._block_loss(ctx.acts, acts_t, B, w["up_block"], uncond_half=True, seed=backward)

    @property
    def need_teacher(self):
        return self.w["block"] > 0 or self.w["dist"] > 0

    def forward_pair(self, teacher_fn, student_fn):
        """Teacher pass and student forward in lockstep on the CURRENT stream: both are recorded (launch wrappers of
        pdm._pdmk defer themselves), then issued side by side with pdm._pdmk.run_lockstep.  Returns (teacher_out, ctx)."""
        with k.Recorder() as rt:
            tout = teacher_fn()
        with k.Recorder() as rs:
            ctx = student_fn()
        with phase("fwd_pair"):
            k.run_lockstep(rt.recs, rs.recs)
        return tout, ctx

    # ------------------------------------------------------------------ eager steps
    def _beside(self, fn):
        """Runs fn() on the teacher stream beside what the caller queues next; returns (result, join)."""
        cur = torch.cuda.current_stream()
        ts = self.teacher_stream
        if ts is None:
            return fn(), (lambda: None)
        ts.wait_stream(cur)
        with torch.cuda.stream(ts):
            out = fn()
        return out, (lambda: cur.wait_stream(ts))

    def main_step(self, latents, noise, timesteps, prompt_embeds, backward=True, input_noise=None, teacher_out=None):
        """latents/noise [B,4,H,W] fp32 (latents already x scaling_factor), timesteps int64 [B], prompt_embeds [B,T,ctx].
        input_noise: the perturbed noise of `input_perturbation` (trainer.py:2416-2417, 2427-2428) - it enters the forward
        process, while the target is formed from the clean `noise`.
        teacher_out: (pred, acts) of `teacher_pass` on the same inputs, computed earlier (the teacher is then not run here).
        Returns the device tensor [diff, dist, block, 0] (float64); total = w_diff*diff + w_block*block + w_dist*dist."""
        if backward:
            self.begin_wt_refresh()
        join = lambda: None
        sfwd = lambda: self.student_forward(latents, noise, timesteps, prompt_embeds, train=backward, input_noise=input_noise)
        if teacher_out is None and self.need_teacher and self.lockstep:
            teacher_out, ctx = self.forward_pair(
                lambda: self.teacher_pass(latents, noise, timesteps, prompt_embeds, input_noise), sfwd)
        else:
            if teacher_out is None and self.need_teacher:
                teacher_out, join = self._beside(lambda: self.teacher_pass(latents, noise, timesteps, prompt_embeds, input_noise))
            ctx = sfwd()
        join()
        self.main_loss_heads(ctx, teacher_out, backward)
        if backward:
            with phase("bwd"):
                self._gscale = self.backward()
        return self.losses

    def upper_step(self, latents, noise, timesteps, prompt_embeds, empty_prompt_embeds, backward=True):
        """Concept-suppression objective: w * mse(student(x_t, c), e_u - (e_c - e_u)) with the teacher's cond / uncond
        predictions computed as ONE batch of 2B."""
        if backward:
            self.begin_wt_refresh()
        tfwd = lambda: self.upper_teacher_pass(latents, noise, timesteps, prompt_embeds, empty_prompt_embeds)
        sfwd = lambda: self.student_forward(latents, noise, timesteps, prompt_embeds, train=backward, want_target=False)
        if self.lockstep:
            tout, ctx = self.forward_pair(tfwd, sfwd)
        else:
            tout, join = self._beside(tfwd)
            ctx = sfwd()
            join()
        self.upper_loss_heads(ctx, tout, backward)
        if backward:
            with phase("bwd"):
                self._gscale = self.backward()
        return self.losses

    def optimizer_step(self, upper=False, max_grad_norm=None):
        opt = self.upper_opt if upper else self.opt
        scale = self._gscale
        if max_grad_norm is not None:
            ss = torch.zeros(1, device=self.dev, dtype=torch.float64)
            k.sumsq(self.student.store.grad, self.student.store.total, ss, 0)
            norm = math.sqrt(float(ss.item())) * scale
            scale *= min(1.0, max_grad_norm / (norm + 1e-6))
        return opt.step(grad_scale=scale, zero_grad=True)

    def total(self, losses, upper=False):
        d, s, b = (float(x) for x in losses[:3].tolist())
        w = self.w
        if upper:
            return w["up_dist"] * s + w["up_block"] * b, 0.0, s, b
        return w["diff"] * d + w["block"] * b + w["dist"] * s, d, s, b


class _CapturedStep:
    """The hipGraphs of one step kind: teacher (own memory pool: it runs beside `fwd`), fwd, and the loss heads + backward
    cut into `len(bwd)` graphs; offs[i] = arena offset from which every gradient is final once bwd[i] has run."""
    __slots__ = ("teacher", "fwd", "loss", "bwd", "offs", "keep")

    def all(self):
        return (([self.teacher] if self.teacher is not None else []) + [self.fwd] +
                ([self.loss] if self.loss is not None else []) + list(self.bwd))


class GraphedBilevel:
    """hipGraph replay of the bilevel iteration.  EVERY captured graph is a single-stream (linear) graph:

         teacher  forward diffusion + frozen teacher forward          -> replayed on the teacher stream
         fwd      forward diffusion + student forward                 -> main stream
         bwd[i]   loss heads + the i-th share of the backward pass    -> main stream, after the teacher stream has joined
                  (prefetch mode: the loss heads are a graph of their own, `loss`, so that the teacher's next pass - which
                  overwrites the outputs they read - can be queued right behind them)

       and everything that runs beside something else (teacher pass, dgrad-copy refresh, the AdamW of a finished share of the
       gradient arena, with world > 1 the bucketed all-reduce in front of it) is ordered BETWEEN the graphs with stream
       waits on the process-wide role streams.  A graph with parallel branches makes hipGraphLaunch (ROCm 7.2) walk
       `GraphExec::parallel_streams_` in `hip::Graph::UpdateStreams`, which reads past the end of that vector when two of
       the executor's own streams share the launch stream's hardware queue ("hipGraphLaunch fault"): graphs
       with max_streams == 1 never enter that code.  Inputs are copied into static buffers; lr / bias corrections live in
       device scalars updated outside the graphs."""

    def __init__(self, stepper, B, C, H, W, T, ctx, segments=12, stream_opt=True, prefetch=None):
        self.st = stepper
        dev = stepper.dev
        self.shape = (B, C, H, W, T, ctx)
        self.lat = torch.zeros(B, C, H, W, device=dev)
        self.noise = torch.zeros(B, C, H, W, device=dev)
        self.t = torch.zeros(B, dtype=torch.int64, device=dev)
        self.ehs = torch.zeros(B, T, ctx, device=dev)
        self.empty = torch.zeros(B, T, ctx, device=dev)
        self.g_main = self.g_upper = None
        # shares of the gradient arena = backward graphs (PDMK_BWD_SEGMENTS: A/B knob).  The AdamW of the LAST share has nothing to
        # hide behind: 12 shares measured +0.5 % over 6 (189.6 -> 190.6 images/s), 18 / 24 / 36 the same as 12
        self.segments = int(os.environ.get("PDMK_BWD_SEGMENTS", segments))
        self.force_segments = False          # tests: cut the backward into segments on a single rank without streamed AdamW
        # AdamW of every finished share of the arena runs beside the rest of the backward (valid without gradient-norm
        # clipping, which needs all gradients first; the shipped configs do not clip: trainer.py:2784-2786)
        self.stream_opt = stream_opt and os.environ.get("PDMK_STREAM_OPT", "1") != "0"     # (0: A/B switch - AdamW after the backward)
        opt_hi = os.environ.get("PDMK_OPT_PRIO", "0") == "1"      # (A/B knob: the streamed AdamW on a high-priority stream)
        self.opt_stream = k.role_stream(dev, "opt_hi" if opt_hi else "opt", high_priority=opt_hi)
        self.cap_stream = k.role_stream(dev, "capture")
        self.closed = False
        # Cross-step teacher prefetch (PDMK_TEACHER_PREFETCH=1, or prefetch=True): the frozen teacher's pass of the NEXT step in
        # program order - the following main batch (`next_batch=, next_id=`) or the upper step that follows this main step
        # (`next_upper=, upper_id=`) - is replayed on the teacher stream as soon as this step's loss heads have read the current
        # outputs, i.e. beside the whole backward, the AdamW and the next student forward, instead of beside its own student
        # forward alone.  The teacher graphs then read their OWN static inputs; a prefetched pass is only used when the caller
        # names the batch again (`batch_id`), else the teacher runs in line as before.  Measured (same box, B = 8):
        # 185.6 -> 190.3 images/s with the main pass alone; queued behind later backward graphs it is worth less and less.
        self.prefetch = ((prefetch if prefetch is not None else os.environ.get("PDMK_TEACHER_PREFETCH", "0") == "1")
                         and stepper.teacher_stream is not None and not stepper.lockstep and stepper.need_teacher)
        self.t_in = ([torch.zeros_like(b) for b in (self.lat, self.noise, self.t, self.ehs)] if self.prefetch else None)
        self.u_in = ([torch.zeros_like(b) for b in (self.lat, self.noise, self.t, self.ehs, self.empty)] if self.prefetch else None)
        self._ahead = {"main": None, "upper": None}      # id of the batch whose teacher outputs are queued / done on the teacher stream
        self.prefetch_hits = 0               # steps that found their teacher pass queued (tests, bench extras)

    def _load(self, lat, noise, t, ehs, empty=None):
        self.lat.copy_(lat); self.noise.copy_(noise); self.t.copy_(t); self.ehs.copy_(ehs)
        if empty is not None:
            self.empty.copy_(empty)

    def _load_teacher(self, upper, *srcs):
        """The teacher graph's own static inputs (prefetch mode), copied on the CURRENT stream."""
        for dst, src in zip(self.u_in if upper else self.t_in, srcs):
            dst.copy_(src)

    def _queue_teacher(self, cs, upper, srcs, token):
        """Queues the teacher pass of a LATER step (inputs `srcs`) on the teacher stream, behind everything queued on the current
        stream so far - the loss heads that read the outputs it will overwrite."""
        ts, cur = self.st.teacher_stream, torch.cuda.current_stream()
        ts.wait_stream(cur)
        with torch.cuda.stream(ts):
            for src in srcs:
                if src.is_cuda:
                    src.record_stream(ts)
            self._load_teacher(upper, *srcs)
            cs.teacher.replay()
        self._ahead["upper" if upper else "main"] = token

    def _claim(self, upper, batch_id, srcs):
        """True when the teacher pass of this step is already queued under `batch_id`; else its inputs are loaded for an in-line pass."""
        kind = "upper" if upper else "main"
        have = self.prefetch and batch_id is not None and self._ahead[kind] == batch_id
        if self.prefetch and not have:
            if self._ahead[kind] is not None:                        # a prefetched pass nobody asked for again is still reading
                torch.cuda.current_stream().wait_stream(self.st.teacher_stream)      # the teacher's inputs
            self._load_teacher(upper, *srcs)
        self._ahead[kind] = None
        self.prefetch_hits += int(have)
        return have

    # ------------------------------------------------------------------ capture
    def capture(self, bilevel=True):
        st = self.st
        st.defer_reduce = True
        # the eager warm-up below really runs the optimiser kernels: snapshot the training state and put it back after
        store = st.student.store
        opts = [o for o in (st.opt, st.upper_opt) if o is not None]
        snap = [store.master.clone(), store.grad.clone()] + [t_.clone() for o in opts for t_ in (o.m, o.v)]
        # eager warm-up (allocator + lazy tables).  It is also where the library times its GEMM candidates for every shape
        # of the step (plan cache), so the static inputs hold random data for it: all-zero operands run at a higher clock
        # and would rank the candidates differently.
        gen = torch.Generator(device=self.lat.device).manual_seed(1234)
        for buf in (self.lat, self.noise, self.ehs, self.empty):
            buf.normal_(generator=gen)
        self.t.random_(0, 1000, generator=gen)
        if self.prefetch:
            self._load_teacher(False, self.lat, self.noise, self.t, self.ehs)
            self._load_teacher(True, self.lat, self.noise, self.t, self.ehs, self.empty)
        cap = self.cap_stream
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            st.main_step(self.lat, self.noise, self.t, self.ehs)
            st.opt.launch(st._gscale)
            if bilevel:
                st.upper_step(self.lat, self.noise, self.t, self.ehs, self.empty)
                st.upper_opt.launch(st._gscale)
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        self.g_main = self._capture_step(upper=False)
        if bilevel:
            self.g_upper = self._capture_step(upper=True)
        torch.cuda.synchronize()
        store.master.copy_(snap[0]); store.grad.copy_(snap[1])
        for i, o in enumerate(opts):
            o.m.copy_(snap[2 + 2 * i]); o.v.copy_(snap[3 + 2 * i])
        del snap
        store.refresh()
        torch.cuda.synchronize()

    def _capture_step(self, upper):
        """Captures one step kind as single-stream graphs.  The tape is cut at block boundaries into `nseg` equal shares of
        the gradient arena (descending offsets): with world > 1 the bucketed all-reduce and then the AdamW of a finished
        share run on the comm stream under the next graph; with world == 1 its AdamW runs on the opt stream."""
        st = self.st
        multi = st.world > 1 or self.force_segments
        nseg = self.segments if (multi or self.stream_opt) else 1
        total = st.student.store.total
        cuts = [total * (nseg - 1 - i) // nseg for i in range(nseg - 1)]      # descending arena offsets
        # (a small LAST share - the one whose AdamW nothing can hide - was measured: 43.3 vs 43.3 ms per main step, no gain)
        cs = _CapturedStep()
        cs.teacher, cs.loss, cs.bwd, cs.offs, cs.keep = None, None, [], [], []
        need_t = upper or st.need_teacher
        cap = self.cap_stream
        state = {"n": 0}

        def seg_cb(off):
            if state["n"] < len(cuts) and off <= cuts[state["n"]]:
                state["n"] += 1
                st.student.engine.flush_pending()      # gradients in [off, total) are final only after this
                cs.bwd[-1].capture_end()
                cs.offs.append(off)
                cs.bwd.append(torch.cuda.CUDAGraph())
                cs.bwd[-1].capture_begin(pool=cs.fwd.pool(), capture_error_mode="thread_local")

        st.segment_cb = seg_cb if nseg > 1 else None
        # like torch.cuda.graph(): collect garbage first, and keep the collector off while capturing - destroying an old
        # CUDAGraph (or freeing its pool) from a GC pass in the middle of a capture aborts the process
        gc.collect()
        torch.cuda.synchronize()
        gc_was_on = gc.isenabled()
        gc.disable()
        cap.wait_stream(torch.cuda.current_stream())
        st.in_graph = True
        try:
            with torch.cuda.stream(cap):
                tout = None
                tin = ((self.u_in if upper else self.t_in) if self.prefetch else
                       (self.lat, self.noise, self.t, self.ehs) + ((self.empty,) if upper else ()))
                tfwd = (lambda: st.upper_teacher_pass(*tin)) if upper else (lambda: st.teacher_pass(*tin))
                sfwd = lambda: st.student_forward(self.lat, self.noise, self.t, self.ehs, train=True, want_target=not upper)
                if need_t and not st.lockstep:             # teacher as its own graph, replayed on the teacher stream
                    cs.teacher = torch.cuda.CUDAGraph()
                    cs.teacher.capture_begin(capture_error_mode="thread_local")
                    tout = tfwd()
                    cs.teacher.capture_end()
                cs.fwd = torch.cuda.CUDAGraph()
                cs.fwd.capture_begin(capture_error_mode="thread_local")
                if need_t and st.lockstep:                 # teacher || student in lockstep inside ONE graph
                    tout, ctx = st.forward_pair(tfwd, sfwd)
                else:
                    ctx = sfwd()
                cs.fwd.capture_end()
                heads = st.upper_loss_heads if upper else st.main_loss_heads
                if self.prefetch and cs.teacher is not None:          # the loss heads as a graph of their own (class docstring)
                    cs.loss = torch.cuda.CUDAGraph()
                    cs.loss.capture_begin(pool=cs.fwd.pool(), capture_error_mode="thread_local")
                    heads(ctx, tout, True)
                    cs.loss.capture_end()
                cs.bwd.append(torch.cuda.CUDAGraph())
                cs.bwd[-1].capture_begin(pool=cs.fwd.pool(), capture_error_mode="thread_local")
                if cs.loss is None:
                    heads(ctx, tout, True)
                st._gscale = st.backward()
                cs.bwd[-1].capture_end()
                cs.offs.append(0)
                cs.keep = [tout, ctx]          # outputs the later graphs read: their storage must stay where it is
        finally:
            st.in_graph = False
            st.segment_cb = None
            if gc_was_on:
                gc.enable()
        torch.cuda.current_stream().wait_stream(cap)
        return cs

    # ------------------------------------------------------------------ replay
    def _replay_step(self, cs, opt=None, have_teacher=False, ahead=None):
        """opt: the optimiser to apply (None = gradients only, the caller applies the optimiser).
        have_teacher: the teacher graph of this step is already queued on the teacher stream (prefetched by the previous step);
        in prefetch mode the caller has been through `_claim` (which loads the teacher graphs' own inputs when it returns False).
        ahead: callable that queues the teacher pass(es) of later steps; called once this step's loss heads are queued."""
        st = self.st
        store = st.student.store
        cur = torch.cuda.current_stream()
        side = st.teacher_stream is not None
        ts = st.teacher_stream if side else cur

        if cs.teacher is not None and not have_teacher:
            if side:
                ts.wait_stream(cur)              # inputs loaded; the previous step's loss heads have read the old outputs
            with torch.cuda.stream(ts):
                cs.teacher.replay()
        st.begin_wt_refresh()                    # dgrad copies of the weights the last optimiser step wrote, beside the forward
        cs.fwd.replay()
        if cs.teacher is not None and side:
            cur.wait_stream(ts)
        st.join_wt_refresh()
        if cs.loss is not None:
            cs.loss.replay()
        if ahead is not None:
            ahead()
        fresh = store.dtype == torch.bfloat16
        streamed = opt is not None and self.stream_opt
        if st.world == 1:
            hi = store.total
            for g, off in zip(cs.bwd, cs.offs):
                g.replay()
                if streamed and len(cs.bwd) > 1 and not self.force_segments:
                    self.opt_stream.wait_stream(cur)             # the share [off, hi) is final
                    with torch.cuda.stream(self.opt_stream):
                        opt.launch_range(off, hi, st._gscale)
                    hi = off
            if streamed:
                if hi > 0:
                    opt.launch_range(0, hi, st._gscale)
                cur.wait_stream(self.opt_stream)
                store.refresh(w_is_fresh=fresh, wt=not store.defer_wt)
            return
        red = st.reducer
        red.begin()
        done = store.total                       # AdamW has been issued for [done, total)
        for g, off in zip(cs.bwd, cs.offs):
            g.replay()
            red.ready_down_to(off)               # comm stream waits for the graph just queued, then reduces its whole buckets
            if streamed and red.stream is not None and red.next_hi < done:
                with torch.cuda.stream(red.stream):      # ... and updates the reduced part behind them
                    opt.launch_range(red.next_hi, done, st._gscale)
                done = red.next_hi
        red.finish()
        if streamed:
            opt.launch_range(0, done, st._gscale)
            store.refresh(w_is_fresh=fresh, wt=not store.defer_wt)

    def _ahead_fn(self, next_batch, next_id, next_upper=None, upper_id=None):
        if not self.prefetch:
            return None
        jobs = []
        if next_upper is not None and upper_id is not None and self.g_upper is not None:      # needed first: queued first
            jobs.append((self.g_upper, True, tuple(next_upper), upper_id))
        if next_batch is not None and next_id is not None:
            jobs.append((self.g_main, False, tuple(next_batch), next_id))
        if not jobs:
            return None
        return lambda: [self._queue_teacher(*j) for j in jobs]

    def main(self, lat, noise, t, ehs, batch_id=None, next_batch=None, next_id=None, next_upper=None, upper_id=None):
        """One main step + its AdamW.  Prefetch mode: `next_batch` = (lat, noise, t, ehs) of the following main step under the token
        `next_id`, and / or `next_upper` = (lat, noise, t, ehs, empty) of the upper step that follows this one under `upper_id`; the
        call that passes the same token as `batch_id` finds its teacher outputs already queued."""
        lr = self.st.opt.prepare()               # lr / bias corrections are read by the AdamW launches of the step
        self._load(lat, noise, t, ehs)
        have = self._claim(False, batch_id, (lat, noise, t, ehs))
        self._replay_step(self.g_main, self.st.opt, have_teacher=have, ahead=self._ahead_fn(next_batch, next_id, next_upper, upper_id))
        if not self.stream_opt:
            self.st.opt.launch(self.st._gscale)
        return lr

    def upper(self, lat, noise, t, ehs, empty, batch_id=None, next_batch=None, next_id=None):
        lr = self.st.upper_opt.prepare()
        self._load(lat, noise, t, ehs, empty)
        have = self._claim(True, batch_id, (lat, noise, t, ehs, empty))
        self._replay_step(self.g_upper, self.st.upper_opt, have_teacher=have, ahead=self._ahead_fn(next_batch, next_id))
        if not self.stream_opt:
            self.st.upper_opt.launch(self.st._gscale)
        return lr

    def close(self):
        """Ordered teardown at a defined idle point: drain the device, destroy the executors (the pool owner `fwd` last),
        release the static buffers.  Trainer calls this when a cached shape is evicted."""
        if self.closed:
            return
        self.closed = True
        torch.cuda.synchronize()
        for cs in (self.g_upper, self.g_main):
            if cs is None:
                continue
            cs.keep = []
            for g in reversed(cs.bwd):
                g.reset()
            cs.bwd = []
            if cs.loss is not None:
                cs.loss.reset()
            if cs.teacher is not None:
                cs.teacher.reset()
            cs.fwd.reset()
        self.g_main = self.g_upper = None
        torch.cuda.synchronize()
