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
        row = self._sched_host[self.t % self._sched_host.shape[0]]      # 32 steps of run-ahead before a row is reused
        row[0], row[1], row[2] = lr, 1 - self.betas[0] ** self.t, 1 - self.betas[1] ** self.t
        self._sched_dev.copy_(row, non_blocking=True)
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
    reduced in `bucket_mb` slices issued on a side stream as soon as the backward pass has produced them (the arena is
    laid out in forward order, so the tail is final first); the division by world size is folded into AdamW."""

    def __init__(self, store, bucket_mb=64):
        self.store = store
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.bucket = bucket_mb * (1 << 20) // 4
        self.stream = torch.cuda.Stream() if (self.world > 1 and store.master.is_cuda) else None
        self.handles = []
        self.next_hi = store.total
        self.flush_cb = None               # set by the stepper: the engine's flush_pending()
        # PDMK_COMM=native: the all-reduces go through the library's own communicator handle (pdmk_comm_t, RCCL bound inside
        # libpdmk.so) instead of torch.distributed's process group; the 128-byte id travels over torch.distributed once.
        # Default: torch.distributed (backend "nccl" = RCCL) - the path the gloo rehearsal tests cover.
        self.comm = None
        if self.world > 1 and store.master.is_cuda and os.environ.get("PDMK_COMM") == "native":
            box = [k.Comm.unique_id() if dist.get_rank() == 0 else None]
            dist.broadcast_object_list(box, src=0)
            self.comm = k.Comm(box[0], dist.get_rank(), self.world)

    def begin(self):
        self.next_hi = self.store.total
        self.handles = []

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

    def _launch(self, lo, hi):
        g = self.store.grad[lo:hi]
        reduce_ = self.comm.all_reduce_sum_ if self.comm is not None else (lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM))
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
            self._launch(0, self.next_hi)
            self.next_hi = 0
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        return 1.0 / self.world


class BilevelStepper:
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
        self.segment_cb = None
        self.after_loss_cb = None      # GraphedBilevel: cut the captured graph between the loss heads and the backward
        self._gscale = 1.0 / world
        # the frozen teacher pass and the student forward are independent until the loss heads: they run on two HIP
        # streams (two parallel branches once captured in a hipGraph) so the small-grid layers of one fill the CUs the
        # other leaves idle
        # teacher forward on its own stream (independent of the student forward); PDMK_TEACHER_STREAM=0 runs it in line
        self.teacher_stream = (torch.cuda.Stream(device=self.dev) if os.environ.get("PDMK_TEACHER_STREAM", "1") != "0"
                               else None)
        self.losses = torch.zeros(4, device=self.dev, dtype=torch.float64)   # 32 bytes: zeroed by k.zero_   # diff, dist, block, (unused)
        # transposed (dgrad) weight copies are refreshed at the START of the next training step, beside its forward, instead
        # of at the end of the optimiser step (PDMK_DEFER_WT=0: refresh with the optimiser as before)
        self.wt_stream = torch.cuda.Stream(device=self.dev)
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
        return e.to(self.dev).to(self.student.dtype).reshape(e.shape[0] * e.shape[1], e.shape[2]).contiguous()

    def _block_loss(self, acts_s, acts_t, B, weight, t_row0=0, seed=True):
        """(1/9) sum_k mse(student_k, teacher_k) (trainer.py:2475-2481) and its gradient seeds."""
        # when every layer of a block is dropped and it has no sampler, two hook keys hold the SAME activation (e.g. both
        # resnets of down_blocks.3 dropped: acts['d3'] is acts['d2']): its gradient seed is the sum of both terms
        seeded = set()
        for key in BLOCK_KEYS:
            a, b = acts_s[key], acts_t[key]
            M, C = a.t.shape
            bt = b.t[t_row0:t_row0 + M]
            n = len(BLOCK_KEYS) * M * C
            again = id(a) in seeded
            if seed and weight > 0 and not again:
                a.g = torch.empty_like(a.t)
                seeded.add(id(a))
            k.mse_fwd_bwd(a.t, bt, None, self.losses, 2, a.g if (seed and weight > 0) else None, B, M // B, C,
                          a.t.stride(0), bt.stride(0), C, 1.0 / n, 2.0 * weight / n, again)

    def _begin_wt_refresh(self):
        """The dgrad copies `wt` (W^T, flipped conv taps) of the weights the last optimiser step wrote are only read by the
        backward pass: their refresh (one 3.4 GB HBM-bound pass) runs on a side stream beside the forward."""
        store = self.student.store
        if not store.defer_wt:
            return
        cur = torch.cuda.current_stream()
        self.wt_stream.wait_stream(cur)
        with torch.cuda.stream(self.wt_stream), phase("wt_refresh"):
            store.refresh_wt()
        self._wt_pending = True

    def _backward_and_reduce(self):
        if self._wt_pending:
            torch.cuda.current_stream().wait_stream(self.wt_stream)
            self._wt_pending = False
        if self.defer_reduce:          # graph mode: the all-reduce is issued by the caller between captured graphs
            self.student.engine.grad_ready_cb = self.segment_cb      # None, or GraphedBilevel's capture-segment switch
            self.student.engine.backward()
            return 1.0 / self.world
        self.reducer.begin()
        self.student.engine.grad_ready_cb = self.reducer.ready_down_to
        self.student.engine.backward()
        return self.reducer.finish()

    def reduce_now(self):
        self.reducer.begin()
        return self.reducer.finish()

    # ------------------------------------------------------------------ steps
    def teacher_pass(self, latents, noise, timesteps, prompt_embeds, input_noise=None):
        """The frozen teacher's part of a main step on its own: forward diffusion + dense forward (trainer.py:2446-2448).
        Returns (pred Act, {block key: Act}).  The teacher does not depend on the student, so GraphedBilevel runs this for
        batch i+1 beside the student's backward pass of batch i and hands the result to main_step(teacher_out=...)."""
        B, C, H, W = latents.shape
        with phase("fwd_teacher"):
            noisy, _ = self._diffuse(latents, noise if input_noise is None else input_noise, timesteps, False)
            return self.teacher.forward_nhwc(noisy, timesteps, self._ehs2d(prompt_embeds), B, H, W, train=False)

    def main_step(self, latents, noise, timesteps, prompt_embeds, backward=True, input_noise=None, teacher_out=None):
        """latents/noise [B,4,H,W] fp32 (latents already x scaling_factor), timesteps int64 [B], prompt_embeds [B,T,ctx].
        input_noise: the perturbed noise of `input_perturbation` (trainer.py:2416-2417, 2427-2428) - it enters the forward
        process, while the target is formed from the clean `noise`.
        teacher_out: (pred, acts) of `teacher_pass` on the same inputs, computed earlier (the teacher is then not run here).
        Returns the device tensor [diff, dist, block, 0] (float64); total = w_diff*diff + w_block*block + w_dist*dist."""
        B, C, H, W = latents.shape
        w = self.w
        need_teacher = w["block"] > 0 or w["dist"] > 0
        if backward:
            self._begin_wt_refresh()
        with phase("diffuse"):
            noisy, target = self._diffuse(latents, noise, timesteps, True)
            if input_noise is not None:
                noisy, _ = self._diffuse(latents, input_noise, timesteps, False)
            ehs = self._ehs2d(prompt_embeds)
            k.zero_(self.losses)
        cur = torch.cuda.current_stream()
        ts = self.teacher_stream if self.teacher_stream is not None else cur
        if teacher_out is not None:
            pred_t, acts_t = teacher_out
        elif need_teacher:
            ts.wait_stream(cur)
            with torch.cuda.stream(ts), phase("fwd_teacher"):
                pred_t, acts_t = self.teacher.forward_nhwc(noisy, timesteps, ehs, B, H, W, train=False)
        with phase("fwd_student"):
            pred, acts = self.student.forward_nhwc(noisy, timesteps, ehs, B, H, W, train=backward)
        if need_teacher and teacher_out is None:
            cur.wait_stream(ts)
        with phase("loss"):
            wb = self.snr_w[timesteps].contiguous()
            HW, cp, n = H * W, pred.t.shape[1], B * H * W * C
            k.mse_fwd(pred.t, target, wb, self.losses, 0, B, HW, C, cp, cp, 1.0 / n)
            if w["dist"] > 0:
                k.mse_fwd(pred.t, pred_t.t, None, self.losses, 1, B, HW, C, cp, cp, 1.0 / n)
            if backward:
                pred.g = k.zeros(tuple(pred.t.shape), pred.t.device, pred.t.dtype)
                k.mse_bwd(pred.t, target, wb, pred.g, B, HW, C, cp, cp, cp, 2.0 * w["diff"] / n, False)
                if w["dist"] > 0:
                    k.mse_bwd(pred.t, pred_t.t, None, pred.g, B, HW, C, cp, cp, cp, 2.0 * w["dist"] / n, True)
            if w["block"] > 0:
                self._block_loss(acts, acts_t, B, w["block"], seed=backward)
        self.last_pred = pred
        if self.after_loss_cb is not None:
            self.after_loss_cb()
        if backward:
            with phase("bwd"):
                self._gscale = self._backward_and_reduce()
        return self.losses

    def upper_step(self, latents, noise, timesteps, prompt_embeds, empty_prompt_embeds, backward=True):
        """Concept-suppression objective: w * mse(student(x_t, c), e_u - (e_c - e_u)) with the teacher's cond / uncond
        predictions computed as ONE batch of 2B."""
        B, C, H, W = latents.shape
        w = self.w
        if backward:
            self._begin_wt_refresh()
        noisy, _ = self._diffuse(latents, noise, timesteps, False)
        ehs = self._ehs2d(prompt_embeds)
        ehs2 = torch.cat([ehs, self._ehs2d(empty_prompt_embeds)], 0)
        noisy2 = torch.cat([noisy, noisy], 0)
        t2 = torch.cat([timesteps, timesteps], 0)
        k.zero_(self.losses)
        cur = torch.cuda.current_stream()
        ts = self.teacher_stream if self.teacher_stream is not None else cur
        ts.wait_stream(cur)
        with torch.cuda.stream(ts), phase("fwd_teacher"):
            pred_t, acts_t = self.teacher.forward_nhwc(noisy2, t2, ehs2, 2 * B, H, W, train=False)
        with phase("fwd_student"):
            pred, acts = self.student.forward_nhwc(noisy, timesteps, ehs, B, H, W, train=backward)
        cur.wait_stream(ts)
        M = B * H * W
        e_c, e_u = pred_t.t[:M], pred_t.t[M:]
        k.axpby(e_c, e_u, -1.0, 2.0)               # e_u <- 2 e_u - e_c  == e_u - (e_c - e_u)
        cp, n = pred.t.shape[1], M * C
        k.mse_fwd(pred.t, e_u, None, self.losses, 1, B, H * W, C, cp, cp, 1.0 / n)
        if backward:
            pred.g = k.zeros(tuple(pred.t.shape), pred.t.device, pred.t.dtype)
            k.mse_bwd(pred.t, e_u, None, pred.g, B, H * W, C, cp, cp, cp, 2.0 * w["up_dist"] / n, False)
        if w["up_block"] > 0:
            # the reference's teacher hooks hold the LAST teacher call (the unconditional one), trainer.py:2951-2954
            self._block_loss_rows(acts, acts_t, B, w["up_block"], backward)
        self.last_pred = pred
        if backward:
            with phase("bwd"):
                self._gscale = self._backward_and_reduce()
        return self.losses

    def _block_loss_rows(self, acts_s, acts_t, B, weight, seed):
        """Upper-step block term against the teacher's UNCONDITIONAL half (rows [M, 2M) of the 2B teacher batch)."""
        seeded = set()
        for key in BLOCK_KEYS:
            a, b = acts_s[key], acts_t[key]
            M, C = a.t.shape
            bt = b.t[M:2 * M]
            n = len(BLOCK_KEYS) * M * C
            again = id(a) in seeded
            if seed and not again:
                a.g = torch.empty_like(a.t)
                seeded.add(id(a))
            k.mse_fwd_bwd(a.t, bt, None, self.losses, 2, a.g if seed else None, B, M // B, C, a.t.stride(0), bt.stride(0), C,
                          1.0 / n, 2.0 * weight / n, again)

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


class GraphedBilevel:
    """hipGraph replay of the bilevel iteration (launch-bound Python loop -> 4 captured graphs):
         g_main  = forward diffusion + teacher fwd + student fwd/bwd + loss heads      (main step)
         g_opt   = fused AdamW + weight-copy refresh                                   (main optimiser)
         g_upper / g_uopt = the same for the concept-suppression step and its optimiser
       Inputs are copied into static buffers; lr / bias corrections live in device scalars updated outside the graphs;
       with world > 1 the bucketed RCCL all-reduce runs eagerly on its side stream between g_main and g_opt."""

    def __init__(self, stepper, B, C, H, W, T, ctx, segments=6, stream_opt=True, prefetch=None, teacher_group=None):
        self.st = stepper
        dev = stepper.dev
        self.lat = torch.zeros(B, C, H, W, device=dev)
        self.noise = torch.zeros(B, C, H, W, device=dev)
        self.t = torch.zeros(B, dtype=torch.int64, device=dev)
        self.ehs = torch.zeros(B, T, ctx, device=dev)
        self.empty = torch.zeros(B, T, ctx, device=dev)
        # Cross-step teacher prefetch (off by default; PDMK_TEACHER_PREFETCH=1 or prefetch=True): the frozen teacher's forward
        # of the NEXT main batch is its own graph, replayed on the teacher stream beside this iteration's student forward AND
        # backward; its outputs are copied into static buffers once this iteration's loss heads have read the previous ones.
        # Measured on one MI355X (same-box A/B): no gain (159.3 img/s without, 157.0 with) - the GEMM kernels
        # of the two streams each fill the CUs' LDS, so they do not co-run and the sum of kernel times is what counts.
        need_teacher = stepper.w["block"] > 0 or stepper.w["dist"] > 0
        self.prefetch = (os.environ.get("PDMK_TEACHER_PREFETCH", "0") == "1" if prefetch is None else prefetch) and \
            need_teacher and stepper.teacher_stream is not None
        # Teacher grouping (PDMK_TEACHER_GROUP=k or teacher_group=k, default 1 = off): the frozen teacher's forward of THIS
        # batch and of the next k-1 announced main batches runs as ONE dense forward over k*B images (its own graph), and each
        # of the k main steps copies its slice of the outputs into the static buffers the captured loss heads read.  The
        # teacher does not depend on the student, so the arithmetic per image is unchanged - but its GEMMs have k times the
        # rows, and at B = 8 the step is bound by what a small GEMM can take in per CU, not by FLOPs.
        g_env = int(os.environ.get("PDMK_TEACHER_GROUP", "1"))
        self.tgroup = max(1, int(teacher_group if teacher_group is not None else g_env)) if (need_teacher and not self.prefetch) else 1
        if self.prefetch or self.tgroup > 1:
            kb = self.tgroup * B
            self.n_lat, self.n_noise = torch.zeros(kb, C, H, W, device=dev), torch.zeros(kb, C, H, W, device=dev)
            self.n_t, self.n_ehs = torch.zeros(kb, dtype=torch.int64, device=dev), torch.zeros(kb, T, ctx, device=dev)
        self._tslots = {}                    # teacher grouping: batch identity -> slot of the last grouped teacher pass
        self.g_teach = self.g_tcopy = None
        self.T_out = None                    # (pred Act, {key: Act}) static teacher outputs read by the captured loss heads
        self._primed = None                  # identity of the batch the static teacher outputs currently belong to
        self.g_main = self.g_opt = self.g_upper = self.g_uopt = None
        self.segments = segments
        self.force_segments = False          # tests: cut the backward into segments on a single rank as well
        # AdamW of every finished sixth of the arena runs beside the rest of the backward (valid without gradient-norm
        # clipping, which needs all gradients first; the shipped configs do not clip: trainer.py:2784-2786)
        self.stream_opt = stream_opt
        self.opt_stream = torch.cuda.Stream(device=dev)

    def _load(self, lat, noise, t, ehs, empty=None):
        self.lat.copy_(lat); self.noise.copy_(noise); self.t.copy_(t); self.ehs.copy_(ehs)
        if empty is not None:
            self.empty.copy_(empty)

    def capture(self, bilevel=True):
        st = self.st
        st.defer_reduce = True
        # the eager warm-up below really runs the optimiser kernels: snapshot the training state and put it back after
        store = st.student.store
        opts = [o for o in (st.opt, st.upper_opt) if o is not None]
        snap = [store.master.clone(), store.grad.clone()] + [t_.clone() for o in opts for t_ in (o.m, o.v)]
        # eager warm-up on a side stream (allocator + lazy tables), as torch.cuda.graphs requires.  It is also where the
        # library times its GEMM candidates for every shape of the step (plan cache), so the static inputs hold random
        # data for it: all-zero operands run at a higher clock and would rank the candidates differently.
        gen = torch.Generator(device=self.lat.device).manual_seed(1234)
        for buf in (self.lat, self.noise, self.ehs, self.empty):
            buf.normal_(generator=gen)
        self.t.random_(0, 1000, generator=gen)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        ext = self.prefetch or self.tgroup > 1          # the teacher runs outside the main step's graph
        if ext:
            G_ = self.tgroup
            for buf, src in ((self.n_lat, self.lat), (self.n_noise, self.noise), (self.n_t, self.t), (self.n_ehs, self.ehs)):
                buf.copy_(src.repeat(G_, *([1] * (src.dim() - 1))))
        with torch.cuda.stream(side):
            if ext:                           # static teacher-output buffers (ONE batch), shaped by one eager pass
                from ..models.unet.engine import Act
                tp, ta = st.teacher_pass(self.n_lat, self.n_noise, self.n_t, self.n_ehs)
                one = lambda a: torch.empty((a.t.shape[0] // self.tgroup, a.t.shape[1]), device=a.t.device, dtype=a.t.dtype)
                self.T_out = (Act(one(tp), rg=False), {k_: Act(one(a), rg=False) for k_, a in ta.items()})
                self._copy_teacher(tp, ta, 0)
                del tp, ta
            st.main_step(self.lat, self.noise, self.t, self.ehs, teacher_out=self.T_out)
            st.opt.launch(st._gscale)
            if bilevel:
                st.upper_step(self.lat, self.noise, self.t, self.ehs, self.empty)
                st.upper_opt.launch(st._gscale)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if ext:
            self._capture_teacher()
        self.g_main, self.main_offs = self._capture_step(
            lambda: st.main_step(self.lat, self.noise, self.t, self.ehs, teacher_out=self.T_out), st.opt, cut_after_loss=self.prefetch)
        self.g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_opt, capture_error_mode="thread_local"):
            st.opt.launch(st._gscale)
        if bilevel:
            self.g_upper, self.upper_offs = self._capture_step(
                lambda: st.upper_step(self.lat, self.noise, self.t, self.ehs, self.empty), st.upper_opt)
            self.g_uopt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_uopt, capture_error_mode="thread_local"):
                st.upper_opt.launch(st._gscale)
        torch.cuda.synchronize()
        store.master.copy_(snap[0]); store.grad.copy_(snap[1])
        for i, o in enumerate(opts):
            o.m.copy_(snap[2 + 2 * i]); o.v.copy_(snap[3 + 2 * i])
        del snap
        store.refresh()
        torch.cuda.synchronize()

    def _copy_teacher(self, pred, acts, slot=0):
        """Slot `slot` (one batch's rows) of the teacher pass's outputs -> the static buffers the loss heads read."""
        dp, da = self.T_out
        for src, dst in [(pred.t, dp.t)] + [(a.t, da[key].t) for key, a in acts.items()]:   # (dense teacher: no aliased keys)
            rows = dst.shape[0]
            k.copy2d(src[slot * rows:(slot + 1) * rows], dst, rows, src.shape[1], src.stride(0), dst.stride(0))

    @staticmethod
    def _batch_id(lat, noise, t, ehs):
        """Identity of a batch for the `nxt` hand-over: same tensors, not modified in place since."""
        return tuple((x.data_ptr(), x._version) for x in (lat, noise, t, ehs))

    def _capture_teacher(self):
        """g_teach = forward diffusion + teacher forward of the NEXT batch (own memory pool: it runs concurrently with the
        student's graphs); g_tcopy = its outputs -> the static buffers the captured loss heads read."""
        st = self.st
        gc.collect()
        torch.cuda.synchronize()
        gc_was_on = gc.isenabled()
        gc.disable()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(cap):
                self.g_teach = torch.cuda.CUDAGraph()
                self.g_teach.capture_begin(capture_error_mode="thread_local")
                self._t_live = st.teacher_pass(self.n_lat, self.n_noise, self.n_t, self.n_ehs)
                self.g_teach.capture_end()
                self.g_tslot = []                # one copy graph per slot of a grouped pass
                for j in range(self.tgroup):
                    gj = torch.cuda.CUDAGraph()
                    gj.capture_begin(capture_error_mode="thread_local")
                    self._copy_teacher(*self._t_live, j)
                    gj.capture_end()
                    self.g_tslot.append(gj)
                self.g_tcopy = self.g_tslot[0]
        finally:
            if gc_was_on:
                gc.enable()
        torch.cuda.current_stream().wait_stream(cap)

    def prime(self, lat, noise, t, ehs):
        """Teacher outputs for a batch that was not announced as `next` by the previous main() (first iteration)."""
        for buf, src in ((self.n_lat, lat), (self.n_noise, noise), (self.n_t, t), (self.n_ehs, ehs)):
            buf.copy_(src)
        self.g_teach.replay()
        self.g_tcopy.replay()
        self._primed = self._batch_id(lat, noise, t, ehs)

    def _capture_step(self, fn, opt, cut_after_loss=False):
        """Captures one step (forward + loss heads + backward [+ streamed AdamW]).  The tape is cut at block boundaries into
        `self.segments` equal shares of the gradient arena (descending offsets).  world == 1: one graph; at every cut the
        AdamW of the finished share is forked onto `opt_stream` (a parallel branch of the graph).  world > 1: one graph per
        share, so that on replay the bucketed all-reduce (and then the AdamW) of a finished share runs on the comm stream
        under the next segment.  Returns ([graphs], [arena offset final after each graph])."""
        st = self.st
        multi = st.world > 1 or self.force_segments
        nseg = self.segments if (multi or self.stream_opt) else 1
        store = st.student.store
        total = store.total
        cuts = [total * (nseg - 1 - i) // nseg for i in range(nseg - 1)]      # descending arena offsets
        graphs, offs = [torch.cuda.CUDAGraph()], []
        cap_stream = torch.cuda.Stream()
        cap_stream.wait_stream(torch.cuda.current_stream())
        state = {"hi": total, "n": 0}

        def fork_adamw(lo):
            self.opt_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.opt_stream):
                opt.launch_range(lo, state["hi"], st._gscale)
            state["hi"] = lo

        def seg_cb(off):
            if state["n"] < len(cuts) and off <= cuts[state["n"]]:
                state["n"] += 1
                st.student.engine.flush_pending()      # gradients in [off, total) are final only after this
                if multi:
                    graphs[-1].capture_end()
                    offs.append(off)
                    graphs.append(torch.cuda.CUDAGraph())
                    graphs[-1].capture_begin(pool=graphs[0].pool(), capture_error_mode="thread_local")
                elif self.stream_opt:
                    fork_adamw(off)

        st.segment_cb = seg_cb if nseg > 1 else None

        def cut():       # teacher prefetch: graph boundary between the loss heads and the backward pass
            if st._wt_pending:             # the dgrad-copy refresh forked at the top of the step joins inside this graph
                torch.cuda.current_stream().wait_stream(st.wt_stream)
                st._wt_pending = False
            graphs[-1].capture_end()
            offs.append(total)
            graphs.append(torch.cuda.CUDAGraph())
            graphs[-1].capture_begin(pool=graphs[0].pool(), capture_error_mode="thread_local")
        st.after_loss_cb = cut if cut_after_loss else None
        # like torch.cuda.graph(): collect garbage first, and keep the collector off while capturing - destroying an old
        # CUDAGraph (or freeing its pool) from a GC pass in the middle of a capture aborts the process
        gc.collect()
        torch.cuda.synchronize()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.stream(cap_stream):
                graphs[0].capture_begin(capture_error_mode="thread_local")
                fn()
                if self.stream_opt and not multi:
                    fork_adamw(0)
                    torch.cuda.current_stream().wait_stream(self.opt_stream)
                    store.refresh(w_is_fresh=store.dtype == torch.bfloat16, wt=not store.defer_wt)
                graphs[-1].capture_end()
        finally:
            if gc_was_on:
                gc.enable()
        offs.append(0)
        st.segment_cb = None
        st.after_loss_cb = None
        torch.cuda.current_stream().wait_stream(cap_stream)
        return graphs, offs

    def _replay_step(self, graphs, offs, opt=None, teach=False):
        """opt: the optimiser to stream (None = gradients only, the caller applies the optimiser).
        teach: graphs[0] ends after the loss heads; the next batch's teacher graph runs on the teacher stream beside ALL of
        `graphs` and publishes its outputs once graphs[0] (the reader of the current ones) has been queued."""
        st = self.st
        store = st.student.store
        cur = torch.cuda.current_stream()
        ts = st.teacher_stream

        def after_first():
            if teach:
                ts.wait_stream(cur)              # the loss heads of this iteration have read the static teacher outputs
                with torch.cuda.stream(ts):
                    self.g_tcopy.replay()
        if teach:
            ts.wait_stream(cur)                  # the next batch's inputs are loaded
            with torch.cuda.stream(ts):
                self.g_teach.replay()
        if st.world == 1:
            multi_forced = self.force_segments
            for i, g in enumerate(graphs):
                g.replay()
                if i == 0:
                    after_first()
            if opt is not None and self.stream_opt and multi_forced:          # forced segments on one rank (tests)
                opt.launch_range(0, store.total, st._gscale)
                store.refresh(w_is_fresh=store.dtype == torch.bfloat16, wt=not store.defer_wt)
            if teach:
                cur.wait_stream(ts)
            return
        red = st.reducer
        red.begin()
        done = store.total                       # AdamW has been issued for [done, total)
        for i, (g, off) in enumerate(zip(graphs, offs)):
            g.replay()
            if i == 0:
                after_first()
            red.ready_down_to(off)               # comm stream waits for the segment just queued, then all-reduces its share
            if opt is not None and self.stream_opt and red.stream is not None and red.next_hi < done:
                with torch.cuda.stream(red.stream):      # ... and updates the reduced part behind it
                    opt.launch_range(red.next_hi, done, st._gscale)
                done = red.next_hi
        red.finish()
        if opt is not None and self.stream_opt:
            opt.launch_range(0, done, st._gscale)
            store.refresh(w_is_fresh=store.dtype == torch.bfloat16, wt=not store.defer_wt)
        if teach:
            cur.wait_stream(ts)

    def _grouped_teacher(self, cur, upcoming):
        """Teacher grouping: make the static teacher outputs hold `cur`'s.  If the last grouped pass did not cover it, run
        one over [cur] + the next tgroup-1 announced batches (short lists are padded with cur)."""
        bid = self._batch_id(*cur)
        if not self._tslots.get(bid):
            group = [cur] + [b for b in (upcoming or [])][: self.tgroup - 1]
            self._tslots = {}
            B = cur[0].shape[0]
            for j in range(self.tgroup):
                b = group[j] if j < len(group) else cur
                for buf, src in zip((self.n_lat, self.n_noise, self.n_t, self.n_ehs), b):
                    buf[j * B:(j + 1) * B].copy_(src)
                if j < len(group):
                    self._tslots.setdefault(self._batch_id(*b), []).append(j)
            self.g_teach.replay()
        # every slot serves ONE step: a batch that comes round again gets a fresh teacher pass (nothing is cached across uses)
        self.g_tslot[self._tslots[bid].pop(0)].replay()

    def main(self, lat, noise, t, ehs, nxt=None):
        """One main step + its AdamW.  nxt = the (lat, noise, t, ehs) of the NEXT main() call - or, with teacher grouping,
        a list of the next tgroup-1 of them.  Teacher prefetch: the next batch's teacher forward runs beside this step;
        without nxt (or on the first call) the teacher outputs of THIS batch are computed up front (`prime`).  Teacher
        grouping: one dense teacher forward serves this and the announced batches."""
        lr = self.st.opt.prepare()               # lr / bias corrections are read by the AdamW launches inside the step
        self._load(lat, noise, t, ehs)
        teach = False
        if self.tgroup > 1:
            ups = None if nxt is None else ([nxt] if torch.is_tensor(nxt[0]) else list(nxt))
            self._grouped_teacher((lat, noise, t, ehs), ups)
        elif self.prefetch:
            if nxt is not None and not torch.is_tensor(nxt[0]):
                nxt = nxt[0]
            if self._primed != self._batch_id(lat, noise, t, ehs):
                self.prime(lat, noise, t, ehs)
            if nxt is not None:
                for buf, src in ((self.n_lat, nxt[0]), (self.n_noise, nxt[1]), (self.n_t, nxt[2]), (self.n_ehs, nxt[3])):
                    buf.copy_(src)
                teach = True
                self._primed = self._batch_id(*nxt)
            else:
                self._primed = None
        self._replay_step(self.g_main, self.main_offs, self.st.opt, teach=teach)
        if not self.stream_opt:
            self.g_opt.replay()
        return lr

    def upper(self, lat, noise, t, ehs, empty):
        lr = self.st.upper_opt.prepare()
        self._load(lat, noise, t, ehs, empty)
        self._replay_step(self.g_upper, self.upper_offs, self.st.upper_opt)
        if not self.stream_opt:
            self.g_uopt.replay()
        return lr
