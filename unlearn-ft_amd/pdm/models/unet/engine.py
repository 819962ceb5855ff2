"""Executor of the pruned / dense SD-2.1 U-Net on libpdmk: explicit forward + hand-written backward (no autograd).

Dataflow = UNet2DConditionModelGated.forward (pdm/models/unet/unet_2d_conditional.py:1417-1728) with the leaves of
SURVEY Appendix B; activations are token-major / NHWC 2-D matrices [B*H*W, C] so conv outputs feed the transformer
GEMMs (and back) without any layout change.  Every op appends its backward closure to a tape; `backward()` replays the
tape in reverse.  Gradients w.r.t. parameters are ACCUMULATED into the fp32 grad arena (zeroed by the fused AdamW).
Gradient fan-in (residuals, skip connections, the shared time embedding) is done in GEMM / norm epilogues
(accumulate flags) or by aliasing a finished gradient buffer - there are no standalone "add" passes on the hot path.
"""
import math
import os

import torch

from ... import _pdmk as k
from .params import ParamStore, kv_layout, temb_layout
from .spec import UNetConfig, padc


class Act:
    """A 2-D activation [rows, cols] (row stride = t.stride(0)) and its gradient (same logical shape).
    pend: a second finished gradient buffer waiting to be added to `g` (the residual branch's, blocks.py:379).  Reading
    `.g` adds it first (one strided accumulate pass); the GroupNorm backward of the same tensor - the usual next writer -
    takes it as an extra addend of its own store instead, so the fan-in normally costs no pass at all."""
    __slots__ = ("t", "_g", "rg", "pend", "src", "cs", "parked")

    def __init__(self, t, rg=True):
        self.t, self._g, self.rg, self.pend = t, None, rg, None
        self.parked = False  # pend is a buffer a deferred (grouped) weight gradient still reads: it must not be updated in place
        self.cs = None       # (accumulator [B, 2, ld], first column, columns covered): per-(image, column) sums of this tensor from
                             # its producers' epilogues (pdmk_gemm_args.colstat) - the GroupNorm that reads it skips its statistics pass
        self.src = None      # a GEGLU output: (pre-activation tensor, the projection's Act) - its consumer's input gradient can be
                             # pushed through GEGLU's backward in the GEMM epilogue (PDMK_EPI_GEGLU_BWD)

    def flush(self):
        if self.pend is not None:
            p_, self.pend = self.pend, None
            if self._g is None:          # parked first gradient and nothing else arrived: it IS the gradient (read-only from here on)
                self._g = p_
                return
            k.copy2d(p_, self._g, p_.shape[0], p_.shape[1], p_.stride(0), self._g.stride(0), accumulate=True)

    @property
    def g(self):
        self.flush()
        return self._g

    @g.setter
    def g(self, v):
        assert self.pend is None
        self._g = v


def _ld(t):
    return t.stride(0)


class UNetEngine:
    def __init__(self, cfg: UNetConfig, blocks, store: ParamStore, dtype):
        self.cfg, self.blocks, self.P, self.dtype = cfg, blocks, store, dtype
        self.dev = store.master.device
        half = cfg.block_out_channels[0] // 2
        # frequency table of Timesteps(dim, flip_sin_to_cos=True, shift=0): built exactly like the reference (fp32 exp)
        self.freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half).to(self.dev)
        self.ws = k.groupnorm_ws(self.dev, 64, cfg.norm_num_groups)      # GN scratch, regrown by groupnorm() if B asks for more
        self.tape = []
        self.train = False
        self.macs = 0
        self.count_macs = False
        self.grad_ready_cb = None      # called with an arena offset: every gradient at or beyond it is final
        self.temb_lay, self.temb_cols = temb_layout(cfg, blocks)
        self.kv_lay, self.kv_cols = kv_layout(blocks)
        # skip k (push order) is concatenated behind an h of cat_ch[k] channels (None: its consumer ResBlock is dropped)
        ups = [r for b in blocks if b.kind == "up" for r in b.resnets]
        self.cat_ch = [None if r.dropped else padc(r.cin - r.skip) for r in reversed(ups)]
        # weight-gradient GEMMs only feed the optimiser, so they could run on a side stream next to the dgrad chain: eager mode
        # only, opt-in (PDMK_WGRAD_ASYNC=1), measured slower since the ring kernels own the LDS (round 4, eager: 181.9 images/s in
        # line and grouped, 177.1 in line one by one, 173.9 on the side stream); never under stream capture - captured graphs are
        # single-stream (bilevel.py GraphedBilevel)
        self.wgrad_async = os.environ.get("PDMK_WGRAD_ASYNC", "0") == "1"
        self.wgrad_stream = None          # the dedicated "wgrad" role stream, created on first use
        self.fuse_geglu = os.environ.get("PDMK_FUSE_GEGLU", "1") != "0"     # A/B switch: 0 = projection + GEGLU as two passes
        # LayerNorm in the prologue of the Linear that reads it (linear(ln=...)): 0 = never, 1 = where it measured as a gain (K <= 320,
        # the 128-row register image: +2 ... +13 us per pair at M = 32 768; the 64-row image of K <= 640 loses 0 ... 20 us to the
        # ring kernels it displaces - tools/ln_fuse_bench.py), 2 = wherever the library takes the pair
        self.fuse_ln = int(os.environ.get("PDMK_FUSE_LN", "1"))
        # "fp8_e4m3" attention precision (UNet2DConditionModelPruned.set_attention_precision): Q / K / V rounded to e4m3fn values
        self.attn_fp8 = False
        self.fuse_geglu_bwd = os.environ.get("PDMK_FUSE_GEGLU_BWD", "1") != "0"   # same for the backward (ff.net.2's input gradient)
        self.defer_fanin = os.environ.get("PDMK_DEFER_FANIN", "1") != "0"   # A/B switch: 0 = residual gradients added at once
        self._keep = []                # operands of in-flight side-stream kernels (freed only after a join)
        # GroupNorm / LayerNorm affine gradients: the second-stage reductions of a whole block run as one launch at the
        # block boundary (PDMK_DEFER_PARTIALS=0: one launch per layer, as before)
        self.partials = k.PartialQueue() if os.environ.get("PDMK_DEFER_PARTIALS", "1") != "0" else None
        # Linear weight gradients: splits store partial slabs, one grouped launch adds them (PDMK_WGRAD_SLABS=0: atomics)
        self.slabs = k.SlabQueue() if os.environ.get("PDMK_WGRAD_SLABS", "1") != "0" else None
        # ... the 3x3 conv weight gradients too (PDMK_WGRAD_SLABS_CONV=1): measured, no gain (43.3 vs 43.6 ms per main step:
        # their slabs are sk x 4-60 MB each), so they keep the atomics
        self.conv_slabs = os.environ.get("PDMK_WGRAD_SLABS_CONV", "0") == "1"
        # upsampler convs as four 2x2 phase convs on the low-resolution image (PDMK_CONV_UP2=0: nearest x2 fused into the
        # 3x3 gather, 2.25 x the multiply-accumulates)
        self.up2 = os.environ.get("PDMK_CONV_UP2", "1") != "0"
        # GroupNorm statistics from the producing GEMM's epilogue (PDMK_GN_EPI=0: a statistics pass per GroupNorm).  The
        # per-(image, column) accumulators of one forward pass live in ONE arena zeroed by one launch at its start
        self.gn_epi = os.environ.get("PDMK_GN_EPI", "1") != "0" and dtype == torch.bfloat16
        # Linear weight gradients of a transformer block: collected during the block's backward and issued as grouped launches at
        # its start marker (k.wgrad_group; PDMK_WGRAD_GROUP=0: one launch per weight, as they are produced)
        self.group_wgrad = os.environ.get("PDMK_WGRAD_GROUP", "1") != "0" and dtype == torch.bfloat16 and self.slabs is not None
        # ... and conv1 / conv2 of a ResBlock the same way: built, measured -0.3 ... +0.2 % for the step in three A/Bs (the grouped conv
        # launches are no faster than the two they replace - 4.13 vs 3.94 ms for the class - and GroupNorm's backward statistics
        # pass, which used to run behind a weight-gradient kernel, pays the dgrad's write-back itself: 0.63 -> 1.0 ms), so it is
        # OFF by default (PDMK_WGRAD_GROUP_CONV=1 turns it on)
        self.group_conv_wgrad = self.group_wgrad and os.environ.get("PDMK_WGRAD_GROUP_CONV", "0") == "1"
        self._wg_items = None
        self._cs_arena, self._cs_off, self._cs_need, self._cs_old = None, 0, 0, []
        self._cs_views, self._cs_cats = {}, {}

    # ------------------------------------------------------------------ helpers
    def _empty(self, rows, cols, dtype=None):
        return torch.empty((rows, cols), device=self.dev, dtype=dtype or self.dtype)

    def _grad_into(self, act, rows, cols, absorb=False):
        """Returns (tensor, accumulate) for writing d(act); with absorb=True (tensor, accumulate, addend): the caller's
        kernel also adds `addend` (a pending residual gradient, or None) in the same store."""
        add = None
        if absorb:
            add, act.pend = act.pend, None
        elif act._g is None and act.pend is not None:
            # a parked gradient met a writer that cannot take an addend: it gets a private copy to accumulate into (the parked
            # buffer itself stays as the deferred weight gradient reads it)
            p_, act.pend = act.pend, None
            act.g = self._empty(rows, cols, act.t.dtype)
            k.copy2d(p_, act._g, rows, cols, _ld(p_), _ld(act._g))
            return act._g, True
        if act._g is None:
            act.g = self._empty(rows, cols, act.t.dtype)
            return (act._g, False, add) if absorb else (act._g, False)
        act.flush()
        return (act._g, True, add) if absorb else (act._g, True)

    def _give(self, act, dy, park=False):
        """act.g += dy where dy is a finished gradient buffer: aliased when act has no gradient yet, else parked as the
        pending addend (folded in by the next GroupNorm / LayerNorm backward of act, or on the next read of act.g).
        park: dy is also the operand of a DEFERRED weight gradient (_wg_items): never aliased as a buffer later kernels accumulate
        into - it waits as the pending addend even when it is the first gradient to arrive."""
        if not act.rg:
            return
        if act._g is None and act.pend is not None:
            # a parked first gradient and now a second finished one (no topology of the shipped recipes does this): the sum gets a
            # buffer of its own - neither finished buffer may be written
            p_, act.pend = act.pend, None
            act.g = self._empty(p_.shape[0], p_.shape[1], act.t.dtype)
            k.copy2d(p_, act._g, p_.shape[0], p_.shape[1], _ld(p_), _ld(act._g))
        if act._g is None and park:
            act.pend = dy
            act.parked = True
        elif act._g is None:
            act.g = dy
        elif self.defer_fanin and dy.dtype == act._g.dtype:
            act.flush()
            act.pend = dy
        else:
            k.copy2d(dy, act._g, dy.shape[0], dy.shape[1], _ld(dy), _ld(act._g), accumulate=True)

    def _wgrad(self, fn, *operands):
        """Launch a weight-gradient kernel.  Side stream: it starts once everything queued on the main stream so far
        (in particular dy) is done; its operands stay referenced until the next join."""
        if not self.wgrad_async or torch.cuda.is_current_stream_capturing():
            fn()
            return
        main = torch.cuda.current_stream()
        if self.wgrad_stream is None:
            self.wgrad_stream = k.role_stream(self.dev, "wgrad")
        self.wgrad_stream.wait_stream(main)
        with torch.cuda.stream(self.wgrad_stream):
            fn()
        self._keep.extend(operands)

    def _join_wgrad(self):
        if self.wgrad_async and self._keep and self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)
            self._keep.clear()

    def _wgrad_fence(self):
        """The gradient buffer a side-stream wgrad is reading is about to be handed on (aliased) and accumulated into
        in place by later main-stream kernels: make the main stream wait for the side stream first."""
        if self.wgrad_async and self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)

    def flush_pending(self):
        """Deferred norm-affine gradient reductions (PartialQueue): after this every gradient the tape has produced so far is
        final in the arena.  Called by whoever consumes gradients mid-backward (bucketed all-reduce, streamed AdamW, graph
        cut) and at the end of backward()."""
        if self._wg_items:          # (a consumer in the middle of a transformer block: what was collected so far goes out now)
            items, self._wg_items = self._wg_items, []
            k.wgrad_group(items, self.slabs)
        if self.partials is not None:
            self.partials.flush()
        if self.slabs is not None:
            self.slabs.flush()

    @staticmethod
    def _splitk(m_out, n_out, red, step):
        tiles = ((m_out + 127) // 128) * ((n_out + 127) // 128)
        nk = max(1, red // step)
        return max(1, min(512 // max(tiles, 1), nk // 16, 64))   # >= 16 K-steps per split (measured sweet spot)

    # ------------------------------------------------------------------ ops
    def linear(self, x, key, bias=None, residual=None, out_f32=False, out=None, geglu=False, cs=None, ln=None):
        """ln: key prefix of a LayerNorm whose output this Linear reads (BasicTransformerBlock norm1/2/3, blocks.py:705-867): x is
        the UN-normalised tensor; where the library takes the pair as one launch (pdmk_gemm_args.ln_gamma: the row-block kernel
        normalises the row block in its registers) the LayerNorm has no pass of its own - the normalised rows and (mean, rstd) are
        written only when a backward pass will read them - else the LayerNorm runs first, as a launch of its own.
        out: optional [M, N] view (any row stride) to write into instead of a fresh tensor (concat buffers).
        geglu: the projection is GEGLU's (blocks.py:44-59; weight rows packed (hidden, gate)-interleaved, params.py): returns
        hidden * gelu(gate) [M, N/2], computed in the GEMM's epilogue where the library has the fused kernel (bf16 ring
        kernels; the pre-activation is then only written when a backward pass will need it), else as a second pass."""
        P = self.P
        k.TAG = key                    # lockstep recording: the same layer op of two models carries the same tag
        e = P.by_key[key + ".weight"]
        Np, Kp = e.shape
        M = x.t.shape[0]
        assert x.t.shape[1] == Kp, f"{key}: input has {x.t.shape[1]} cols, weight expects {Kp}"
        y = out if out is not None else self._empty(M, Np, torch.float32 if out_f32 else None)
        assert tuple(y.shape) == (M, Np)
        # time-embedding MLP / batched time_emb_proj: M = batch rows -> weight-streaming kernels (skinny operand in LDS)
        skinny = M <= 16 and residual is None and (8 if M <= 8 else 16) * Kp * 4 + 512 <= 65536
        skinny_dgrad = skinny and (8 if M <= 8 else 16) * Np * 4 + 512 <= 65536
        a_t, ln_args, ln_src = x.t, None, None             # forward A operand; the LayerNorm prologue of the GEMM, its input Act
        if ln is not None:
            take = (self.fuse_ln and self.dtype == torch.bfloat16 and not skinny and residual is None and out is None and
                    not out_f32 and cs is None and (self.fuse_ln >= 2 or Kp <= 320) and
                    k.gemm_ln_supported(x.t, P.wv(key + ".weight"), M, Np, Kp, _ld(x.t), Kp, geglu=geglu, bias=bool(bias)))
            if not take:
                x = self.layernorm(x, ln)
                a_t = x.t
            else:
                ln_src = x
                st_ = torch.empty((M, 2), device=self.dev, dtype=torch.float32) if self.train else None
                lno = self._empty(M, Kp) if self.train else None
                ln_args = (P.p(ln + ".weight"), P.p(ln + ".bias"), st_, lno, 1e-5)
                x = Act(lno)                                 # what the backward pass sees as this Linear's input
                if self.train:
                    lnw, src = P.p(ln + ".weight"), ln_src

                    def lnbwd():                             # runs AFTER this Linear's backward (appended before it)
                        dx, acc_, add = self._grad_into(src, M, Kp, absorb=True)
                        k.layernorm_bwd(src.t, x.g, dx, lnw, st_, P.g(ln + ".weight"), P.g(ln + ".bias"), M, Kp, _ld(src.t),
                                        _ld(x.g), _ld(dx), acc_, queue=self.partials, add=add)
                    self.tape.append(lnbwd)
        gl = None
        acc, acc_ok = None, False       # GroupNorm statistics out of this GEMM's epilogue (cs): the accumulator slice, and whether it was fed
        if geglu:
            assert residual is None and out is None and not out_f32
            gl = self._empty(M, Np // 2)
            fused = (self.fuse_geglu and not skinny and self.dtype == torch.bfloat16 and
                     (ln_args is not None or k.splitk_plan(a_t, P.wv(key + ".weight"), M, Np, Kp, _ld(a_t), Kp) == 1))
            if fused:
                if not self.train:
                    y = None                    # inference (teacher): the [M, N] pre-activation never reaches memory
                fused = k.gemm_geglu(a_t, P.wv(key + ".weight"), gl, y, M, Np, Kp, _ld(a_t), Kp,
                                     bias=P.p(bias) if bias else None, macs=M * e.logical[0] * e.logical[1], ln=ln_args)
                assert fused or ln_args is None, "pdmk_gemm_ln_supported said yes"
                if not fused and y is None:
                    y = self._empty(M, Np)
        if geglu and fused:
            pass
        elif ln_args is not None:      # LayerNorm prologue: one kernel family, no plan to make
            k.gemm(a_t, P.wv(key + ".weight"), y, M, Np, Kp, _ld(a_t), Kp, _ld(y), bias=P.p(bias) if bias else None,
                   macs=M * e.logical[0] * e.logical[1], ln=ln_args)
        elif skinny:
            k.skinny_gemm(x.t, P.wv(key + ".weight"), y, M, Np, Kp, _ld(x.t), Kp, _ld(y), bias=P.p(bias) if bias else None)
        else:
            # cs = (B, rows per image): a GroupNorm reads this output next - its statistics come out of this epilogue
            acc = self._cs_for(y, cs[0], cs[1], M, Np, view=out is not None) if (cs is not None and not out_f32) else None
            got = (k.gemm if out_f32 else k.gemm_auto)(
                x.t, P.wv(key + ".weight"), y, M, Np, Kp, _ld(x.t), Kp, _ld(y), bias=P.p(bias) if bias else None,
                R=residual.t if residual else None, ldr=_ld(residual.t) if residual else 0,
                macs=M * e.logical[0] * e.logical[1],
                **({"out_f32": True} if out_f32 else ({"colstat": acc, "rows_per_b": cs[1]} if acc is not None else {})))
            acc_ok = acc is not None and bool(got)
        lmacs = M * e.logical[0] * e.logical[1]
        if self.count_macs:
            self.macs += lmacs
        if geglu and not fused:
            k.geglu_fwd(y, gl, M, Np // 2, _ld(y), Np // 2, layout=1)
        out = Act(y)
        if acc_ok:
            out.cs = (acc[0], acc[1], Np)
        if self.train:
            def bwd():
                dy = out.g
                if skinny:
                    xt = x.t
                    k.skinny_wgrad(dy, xt, P.g(key + ".weight"), P.g(bias) if bias else None, M, Np, Kp, _ld(dy),
                                   _ld(xt), Kp)
                    if x.rg and skinny_dgrad:
                        dx, acc = self._grad_into(x, M, Kp)
                        k.skinny_gemm(dy, P.wtv(key + ".weight"), dx, M, Kp, Np, _ld(dy), Np, _ld(dx), accumulate=acc)
                    elif x.rg:            # wide projection (all time_emb_proj at once): dy does not fit LDS -> tiled GEMM
                        dyc = dy
                        if dy.dtype != self.dtype:
                            dyc = self._empty(M, Np)
                            k.cast_permute(dy, dyc, M * Np, 1, 1, 0)
                        dx, acc = self._grad_into(x, M, Kp)
                        k.gemm_auto(dyc, P.wtv(key + ".weight"), dx, M, Kp, Np, Np, Np, _ld(dx), accumulate=acc, macs=lmacs)
                    return
                if dy.dtype != self.dtype:        # fp32 output (time-embedding projections): tiny cast for the GEMMs
                    dyc = self._empty(M, Np)
                    k.cast_permute(dy, dyc, M * Np, 1, 1, 0)
                    dy = dyc
                xt = x.t
                collect = self._wg_items is not None and not self.wgrad_async
                if collect:      # inside a transformer block: the weight gradient joins the block's grouped launch (dy and xt
                    # stay referenced - and unmodified, see _give(park=True) - until _wg_flush)
                    self._wg_items.append((dy, xt, P.g(key + ".weight"), Np, Kp, M, _ld(dy), _ld(xt),
                                           P.g(bias) if bias else None, lmacs))
                else:
                    # wgrad first (side stream if enabled), dgrad second: the two GEMMs of one layer can run side by side
                    self._wgrad(lambda: k.wgrad(dy, xt, P.g(key + ".weight"), Np, Kp, M, _ld(dy), _ld(xt), macs=lmacs,
                                                colsum_out=P.g(bias) if bias else None,   # bias gradient fused in
                                                queue=None if self.wgrad_async else self.slabs),
                                dy, xt)
                fused_g = False
                if (x.rg and x.src is not None and x._g is None and self.fuse_geglu_bwd and self.dtype == torch.bfloat16 and
                        k.splitk_plan(dy, P.wtv(key + ".weight"), M, Kp, Np, _ld(dy), Np) == 1):
                    pre, proj = x.src          # gradient of the GEGLU pre-activation straight from this GEMM's epilogue
                    dpre = self._empty(M, 2 * Kp)
                    fused_g = k.gemm_geglu_bwd(dy, P.wtv(key + ".weight"), pre, dpre, M, Kp, Np, _ld(dy), Np, macs=lmacs)
                    if fused_g:
                        proj.g = dpre
                if x.rg and not fused_g:
                    dx, acc = self._grad_into(x, M, Kp)
                    k.gemm_auto(dy, P.wtv(key + ".weight"), dx, M, Kp, Np, _ld(dy), Np, _ld(dx), accumulate=acc,
                                macs=lmacs)
                if residual is not None:
                    self._wgrad_fence()
                    self._give(residual, out.g, park=collect)
            self.tape.append(bwd)
        if geglu:
            act = Act(gl)
            if self.train:
                act.src = (y, out)

                def gbwd():               # runs BEFORE the projection's own backward: d(pre-activation) from d(gl)
                    if out._g is not None:        # already produced by the consumer's fused epilogue (PDMK_EPI_GEGLU_BWD)
                        return
                    out.g = self._empty(M, Np)
                    k.geglu_bwd(y, act.g, out.g, M, Np // 2, _ld(y), _ld(act.g), Np, layout=1)
                self.tape.append(gbwd)
            return act
        return out

    def conv3(self, x, key, B, Hi, Wi, mode, bias, rowvec=None, residual=None, rv_cols=None, out=None, cs=False):
        """3x3 conv, pad 1.  mode 0: stride 1; 1: stride 2; 2: nearest-x2 upsample fused into the gather; 4: stride 2
        padded on the bottom/right only (VAE encoder downsample; forward only).
        rowvec (+ rv_cols = (first column, width)): per-image row added to every pixel = this ResBlock's column slice of
        the batched time-embedding projection [B, sum of widths] (fp32)."""
        P = self.P
        k.TAG = key
        e = P.by_key[key + ".weight"]
        Cop, _, Cip = e.shape
        assert x.t.shape[1] == Cip, f"{key}: input has {x.t.shape[1]} channels, weight expects {Cip}"
        Ho, Wo = ((Hi + 1) // 2, (Wi + 1) // 2) if mode in (1, 4) else ((2 * Hi, 2 * Wi) if mode == 2 else (Hi, Wi))
        M = B * Ho * Wo
        y = out if out is not None else self._empty(M, Cop)
        assert tuple(y.shape) == (M, Cop)
        if (mode == 2 and getattr(self, 'up2', False) and rowvec is None and residual is None and
                k.conv_up2_supported(B, Hi, Wi, Cip, Cop, self.dtype)):
            return self._conv_up2(x, key, B, Hi, Wi, bias, y, e, cs and ("view" if out is not None else "own")), Ho, Wo
        acc = self._cs_for(y, B, Ho * Wo, M, Cop, view=out is not None) if cs else None   # a GroupNorm reads this output next
        acc_ok = k.gemm_auto(x.t, P.wv(key + ".weight"), y, M, Cop, 9 * Cip, 0, 9 * Cip, _ld(y), a_mode=k.A_CONV,
               conv=(B, Hi, Wi, Cip, Ho, Wo, mode, _ld(x.t)), bias=P.p(bias),
               rowvec=rowvec.t[:, rv_cols[0]:] if rowvec is not None else None, rows_per_b=Ho * Wo,
               ldrv=_ld(rowvec.t) if rowvec is not None else 0,
               R=residual.t if residual else None, ldr=_ld(residual.t) if residual else 0,
               macs=M * e.logical[0] * e.logical[1] * 9, colstat=acc)
        lmacs = M * e.logical[0] * e.logical[1] * 9
        if self.count_macs:
            self.macs += lmacs
        out = Act(y)
        if acc is not None and acc_ok:
            out.cs = (acc[0], acc[1], Cop)
        if self.train:
            def bwd():
                dy = out.g
                ldy = _ld(dy)
                xt = x.t
                collect = self._wg_items is not None and self.group_conv_wgrad and not self.wgrad_async and mode == 0
                if collect:      # conv1 / conv2 of a ResBlock: one grouped launch at the block's start marker
                    self._wg_items.append((dy, xt, P.g(key + ".weight"), Cop, 9 * Cip, M, ldy, 0, P.g(bias), lmacs, k.B_COLK_CONV,
                                           (B, Hi, Wi, Cip, Ho, Wo, mode, _ld(xt))))
                else:
                    self._wgrad(lambda: k.wgrad(dy, xt, P.g(key + ".weight"), Cop, 9 * Cip, M, ldy, 0, b_mode=k.B_COLK_CONV,
                                                conv=(B, Hi, Wi, Cip, Ho, Wo, mode, _ld(xt)), macs=lmacs,
                                                colsum_out=P.g(bias),       # bias gradient fused into the weight gradient
                                                queue=None if (self.wgrad_async or not self.conv_slabs) else self.slabs),
                                dy, xt)
                if x.rg:
                    if mode == 2:
                        tmp = self._empty(M, Cip)
                        k.gemm_auto(dy, P.wtv(key + ".weight"), tmp, M, Cip, 9 * Cop, 0, 9 * Cop, Cip, a_mode=k.A_CONV,
                               conv=(B, Ho, Wo, Cop, Ho, Wo, 0, ldy), macs=lmacs)
                        pooled = self._empty(B * Hi * Wi, Cip)
                        k.pool2x2_sum(tmp, pooled, B, Hi, Wi, Cip)
                        self._give(x, pooled)
                    else:
                        dx, acc = self._grad_into(x, B * Hi * Wi, Cip)
                        k.gemm_auto(dy, P.wtv(key + ".weight"), dx, B * Hi * Wi, Cip, 9 * Cop, 0, 9 * Cop, _ld(dx),
                                    a_mode=k.A_CONV, conv=(B, Ho, Wo, Cop, Hi, Wi, 3 if mode == 1 else 0, ldy),
                               accumulate=acc, macs=lmacs)
                if rowvec is not None:
                    # d(rowvec)[b] = column sums of dy over the pixels of image b, written into this block's column slice
                    # of the batched projection's gradient (the conv bias gradient - their sum over b - comes out of the weight
                    # gradient kernel)
                    if rowvec.g is None:
                        rowvec.g = k.zeros(tuple(rowvec.t.shape), self.dev, torch.float32)
                    dtp = rowvec.g[:, rv_cols[0]:]
                    hw = Ho * Wo
                    # (accumulate: the slice was zeroed with the whole gradient above and is written once per backward pass - no
                    # zero-fill launch per ResBlock)
                    k.colsum(dy, dtp, hw, Cop, ldy, accumulate=True, nbatch=B, ldo=_ld(rowvec.g))
                if residual is not None:
                    self._wgrad_fence()
                    self._give(residual, dy, park=collect)
            self.tape.append(bwd)
        return out, Ho, Wo

    def _conv_up2(self, x, key, B, Hi, Wi, bias, y, e, cs=False):
        """Upsample2D (nearest x2 + 3x3 conv; unet_2d_conditional.py up blocks) as four 2x2 phase convs
        on the low-resolution image (pdmk.h conv_mode 5..12): 16 instead of 36 multiply-accumulates per low-resolution pixel,
        forward, input gradient and weight gradient alike.  The four phases of the forward and of the weight gradient are
        independent problems of one shape: one grouped launch each (pdmk_gemm_group)."""
        P = self.P
        Cop, _, Cip = e.shape
        Ml = B * Hi * Wi
        wp, wpt = P.up2_weights(key)
        lmacs = 4 * Ml * e.logical[0] * e.logical[1] * 4           # executed multiply-accumulates (the 3x3 form: 9 / 4 of it)
        geo = lambda m, ci, ld: (B, Hi, Wi, ci, Hi, Wi, m, ld)
        # cs: False = no GroupNorm reads this output; "own" = y is a fresh tensor; "view" = y is a concat view (the four phases add
        # their column sums to that buffer's GroupNorm accumulator; rows are counted on the low-resolution grid a phase enumerates)
        acc = self._cs_for(y, B, Hi * Wi, Ml, Cop, view=cs == "view") if cs else None
        with k.Recorder() as r:
            for p_ in range(4):
                k.gemm(x.t, wp[p_], y, Ml, Cop, 4 * Cip, 0, 4 * Cip, _ld(y), a_mode=k.A_CONV, conv=geo(5 + p_, Cip, _ld(x.t)),
                       bias=P.p(bias), macs=lmacs // 4, colstat=acc, rows_per_b=Hi * Wi if acc is not None else 0)
        self._issue(r.recs)
        if self.count_macs:
            self.macs += 4 * Ml * e.logical[0] * e.logical[1] * 9  # model MACs are counted as the reference executes them
        out = Act(y)
        if acc is not None:
            out.cs = (acc[0], acc[1], Cop)
        if self.train:
            def bwd():
                dy = out.g
                ldy = _ld(dy)
                xt = x.t
                # small low-resolution grids (8x8 -> 16x16 at B = 8: 512 pixels) leave the phase forms of the two gradients with
                # too few workgroups: there the 3x3 forms at the high resolution stay (same arithmetic, measured faster -
                # tools/up2_bench.py)
                # (one MI355X, B = 8, us: weight gradient 32->64 329 -> 262, 16->32 329 -> 279, 8->16 98 -> 145; input gradient
                # 32->64 221 -> 134, 16->32 207 -> 204, 8->16 76 -> 203; forward 301 -> 139, 248 -> 123, 80 -> 55)
                phase_w, phase_d = Ml >= 2048, Ml >= 8192
                if phase_w:
                    # weight gradient: four phase problems into a zeroed [4][Co][4 Ci] buffer, then folded into the 3x3 gradient
                    dwp = k.zeros((4, Cop, 4 * Cip), self.dev, torch.float32)
                    sk = k.wgrad_plan(dy, xt, Cop, 4 * Cip, Ml, ldy, 0, k.B_COLK_CONV, geo(5, Cip, _ld(xt)))
                    with k.Recorder() as rw:
                        for p_ in range(4):
                            k.gemm(dy, xt, dwp[p_], Cop, 4 * Cip, Ml, ldy, 0, 4 * Cip, a_mode=k.A_COLK, b_mode=k.B_COLK_CONV,
                                   conv=geo(5 + p_, Cip, _ld(xt)), out_f32=True, splitk=sk, accumulate=(sk == 1),
                                   dtype=k.dt(xt), colsum_out=P.g(bias), macs=lmacs // 4)
                    self._issue(rw.recs)
                    k.up2_combine_wgrad(dwp, P.g(key + ".weight"), Cop, Cip)
                else:
                    k.wgrad(dy, xt, P.g(key + ".weight"), Cop, 9 * Cip, 4 * Ml, ldy, 0, b_mode=k.B_COLK_CONV,
                            conv=(B, Hi, Wi, Cip, 2 * Hi, 2 * Wi, 2, _ld(xt)), macs=lmacs * 9 // 4, colsum_out=P.g(bias))
                if x.rg and phase_d:
                    # input gradient: all four phases as ONE problem (conv_mode 13: K = (phase, tap, channel))
                    dx, acc = self._grad_into(x, Ml, Cip)
                    k.gemm(dy, wpt, dx, Ml, Cip, 16 * Cop, 0, 16 * Cop, _ld(dx), a_mode=k.A_CONV, conv=geo(13, Cop, ldy),
                           accumulate=acc, macs=lmacs)
                elif x.rg:
                    tmp = self._empty(4 * Ml, Cip)
                    k.gemm_auto(dy, P.wtv(key + ".weight"), tmp, 4 * Ml, Cip, 9 * Cop, 0, 9 * Cop, Cip, a_mode=k.A_CONV,
                                conv=(B, 2 * Hi, 2 * Wi, Cop, 2 * Hi, 2 * Wi, 0, ldy), macs=lmacs * 9 // 4)
                    pooled = self._empty(Ml, Cip)
                    k.pool2x2_sum(tmp, pooled, B, Hi, Wi, Cip)
                    self._give(x, pooled)
            self.tape.append(bwd)
        return out

    def _issue(self, recs):
        """Independent GEMM records of one shape: one grouped launch (or, while an outer lockstep recording is active, handed
        on to it one by one)."""
        if k.RECORD is not None:
            k.RECORD.extend(recs)
        else:
            k.gemm_group(recs)

    def _cs_begin(self):
        """Start of a forward pass: one zero fill for every GroupNorm accumulator of the pass."""
        self.gn_count = [0, 0]
        self.gn_miss = [] if os.environ.get("PDMK_GN_EPI_DEBUG") else None    # (layer, rows, columns) of GroupNorms without them
        if not self.gn_epi:
            return
        # sized for the LARGEST pass seen so far (the teacher alternates B and 2B passes): after one eager pass of every shape -
        # the warm-up that precedes any capture - the arena never grows again, so no capture allocates or zero-fills piecemeal
        self._cs_max = max(getattr(self, "_cs_max", 0), self._cs_need)
        if self._cs_max and (self._cs_arena is None or self._cs_arena.numel() < self._cs_max):
            # grow: the old arena is RETAINED - a captured graph of an earlier (smaller) pass still zeroes and adds into it
            self._cs_old.append(self._cs_arena)
            self._cs_arena = torch.empty(max(self._cs_max, 1 << 16), device=self.dev, dtype=torch.int64)
        self._cs_off, self._cs_need = 0, 0
        self._cs_views, self._cs_cats = {}, {}
        self.gn_count = [0, 0]          # GroupNorms of this pass, of which with statistics from a producer's epilogue
        if self._cs_arena is not None:
            k.zero_(self._cs_arena)

    def _cs_alloc(self, B, cols):
        """Zeroed accumulator [B, 4, cols] (int64 limbs, pdmk.h colstat; a slice of the pass's arena; the first pass of a new shape sizes the arena and
        zeroes its accumulators one by one)."""
        n = B * 4 * cols
        self._cs_need += n
        if self._cs_arena is not None and self._cs_off + n <= self._cs_arena.numel():
            t = self._cs_arena[self._cs_off:self._cs_off + n].view(B, 4, cols)
            self._cs_off += n
            return t
        return k.zeros((B, 4, cols), self.dev, torch.int64)

    @staticmethod
    def _cs_shape_ok(M, B, ld):
        return B > 0 and M % B == 0 and (M // B) % 64 == 0 and ld % 8 == 0

    def _cs_for(self, y, B, rows_per_b, M, N, view=False):
        """(accumulator [B, 4, ld], first column) for a GEMM that writes `y` [M, N] and whose output a GroupNorm reads next -
        or None when the statistics epilogue cannot take the shape (the GroupNorm then runs its own statistics pass).
        view: y is a concat-buffer view handed out by _skip_view / _left_view - the sums go to that buffer's accumulator, at the
        view's columns, so that the GroupNorm over the whole concat finds both halves in one place."""
        if not (self.gn_epi and rows_per_b > 0 and rows_per_b % 64 == 0 and M % 64 == 0 and N % 8 == 0 and
                y.stride(0) % 8 == 0 and y.dtype == torch.bfloat16):
            return None
        if view:
            # keyed by the view's address, and checked against the concat buffer it was registered for (which the entry keeps
            # alive, so the address cannot be recycled inside the pass): same rows, same row stride, columns inside the buffer
            ent = self._cs_views.get(y.data_ptr())
            if (ent is not None and ent[0].shape[0] == B and ent[1] + N <= ent[0].shape[2] and ent[2].shape[0] == y.shape[0] and
                    ent[2].stride(0) == y.stride(0)):
                return ent[0], ent[1]
        return self._cs_alloc(B, N), 0

    def groupnorm(self, x, key, B, HW, G, gs, eps, silu):
        P = self.P
        k.TAG = key
        C = x.t.shape[1]
        y = self._empty(B * HW, C)
        stats = torch.empty((B, G, 2), device=self.dev, dtype=torch.float32)
        gw, gb = P.p(key + ".weight"), P.p(key + ".bias")
        cs = x.cs
        self.gn_count[0] += 1
        if cs is None and self.gn_miss is not None:
            self.gn_miss.append((key, B * HW, C))
        if cs is not None and cs[2] >= G * gs and cs[0].shape[0] == B:
            self.gn_count[1] += 1
            k.groupnorm_apply_colstat(x.t, y, gw, gb, stats, cs[0], cs[1], B, HW, C, _ld(x.t), C, G, gs, eps, silu)
        else:
            self.ws = k.groupnorm_ws(self.dev, B, G, self.ws)
            k.groupnorm_fwd(x.t, y, gw, gb, stats, self.ws, B, HW, C, _ld(x.t), C, G, gs, eps, silu)
        out = Act(y)
        if self.train:
            def bwd():
                dy = out.g
                dx, acc, add = self._grad_into(x, B * HW, C, absorb=True)
                k.groupnorm_bwd(x.t, dy, dx, gw, gb, stats, P.g(key + ".weight"), P.g(key + ".bias"), self.ws, B,
                                HW, C, _ld(x.t), _ld(dy), _ld(dx), G, gs, silu, acc, add=add, queue=self.partials)
            self.tape.append(bwd)
        return out

    def layernorm(self, x, key):
        P = self.P
        k.TAG = key
        M, C = x.t.shape
        y = self._empty(M, C)
        stats = torch.empty((M, 2), device=self.dev, dtype=torch.float32)
        gw = P.p(key + ".weight")
        k.layernorm_fwd(x.t, y, gw, P.p(key + ".bias"), stats, M, C, _ld(x.t), C, 1e-5)
        out = Act(y)
        if self.train:
            def bwd():
                dx, acc, add = self._grad_into(x, M, C, absorb=True)     # add: the residual branch's finished gradient
                k.layernorm_bwd(x.t, out.g, dx, gw, stats, P.g(key + ".weight"), P.g(key + ".bias"), M, C, _ld(x.t),
                                _ld(out.g), _ld(dx), acc, queue=self.partials, add=add)
            self.tape.append(bwd)
        return out

    def attention(self, q, kk, v, B, H, Nq, Nk, q_act, kv_act, q_cols, kv_cols, tag="attn"):
        """q/kk/v: 2-D views [B*N, H*64] (column slices of the projection outputs); *_act own the gradients;
        q_cols / kv_cols = (lo, hi) column ranges of q in q_act and of (k, v) in kv_act."""
        k.TAG = tag
        d = H * 64
        o = self._empty(B * Nq, d)
        lse = torch.empty((B, H, Nq), device=self.dev, dtype=torch.float32)
        qs, ks, vs = (Nq * _ld(q), _ld(q)), (Nk * _ld(kk), _ld(kk)), (Nk * _ld(v), _ld(v))
        os_ = (Nq * d, d)
        scale = 64 ** -0.5
        k.attn_fwd(q, kk, v, o, lse, B, H, Nq, Nk, qs, ks, vs, os_, scale)
        if self.count_macs:
            self.macs += 2 * B * H * Nq * Nk * 64
        out = Act(o)
        if self.train:
            def bwd():
                delta = torch.empty((B, H, Nq), device=self.dev, dtype=torch.float32)
                if q_act.g is None:
                    q_act.g = self._empty(*q_act.t.shape)
                if kv_act.rg and kv_act.g is None:
                    kv_act.g = self._empty(*kv_act.t.shape)
                dq = q_act.g[:, q_cols[0]:q_cols[1]]
                if kv_act.rg:
                    dk = kv_act.g[:, kv_cols[0][0]:kv_cols[0][1]]
                    dv = kv_act.g[:, kv_cols[1][0]:kv_cols[1][1]]
                else:      # text conditioning does not need gradients, but the kernel writes dK/dV: scratch
                    scratch = self._empty(B * Nk, 2 * d)
                    dk, dv = scratch[:, :d], scratch[:, d:]
                k.attn_bwd(q, kk, v, o, out.g, lse, delta, dq, dk, dv, B, H, Nq, Nk, qs, ks, vs, os_,
                           (Nq * _ld(dq), _ld(dq)), (Nk * _ld(dk), _ld(dk)), (Nk * _ld(dv), _ld(dv)), scale)
            self.tape.append(bwd)
        return out

    def geglu(self, x):
        M, F2 = x.t.shape
        Fd = F2 // 2
        y = self._empty(M, Fd)
        k.geglu_fwd(x.t, y, M, Fd, _ld(x.t), Fd)
        out = Act(y)
        if self.train:
            def bwd():
                assert x.g is None
                x.g = self._empty(M, F2)
                k.geglu_bwd(x.t, out.g, x.g, M, Fd, _ld(x.t), _ld(out.g), F2)
            self.tape.append(bwd)
        return out

    def silu(self, x):
        y = torch.empty_like(x.t)
        k.silu_fwd(x.t, y)
        out = Act(y)
        if self.train:
            def bwd():
                assert x.g is None and out.g.is_contiguous()
                x.g = torch.empty_like(x.t)
                k.silu_bwd(x.t, out.g, x.g)
            self.tape.append(bwd)
        return out

    def concat(self, a, b, cat=None):
        """torch.cat([a, b], dim=1) of the up path.  `cat` = the [M, Ca + Cb] buffer whose right columns the
        skip b is a view of (its producer wrote them in place); a was normally written into the left columns by ITS producer
        (`out=`), so no copy happens at all; whatever is not in place yet is copied in."""
        M, Ca, Cb = a.t.shape[0], a.t.shape[1], b.t.shape[1]
        if cat is not None and tuple(cat.shape) != (M, Ca + Cb):
            cat = None
        inplace = cat is not None
        if cat is None:
            cat = self._empty(M, Ca + Cb)
            k.copy2d(b.t, cat[:, Ca:], M, Cb, _ld(b.t), Ca + Cb)
        if not (a.t.data_ptr() == cat.data_ptr() and _ld(a.t) == Ca + Cb):
            k.copy2d(a.t, cat, M, Ca, _ld(a.t), Ca + Cb)
            inplace = False
        out = Act(cat)
        acc = self._cs_cats.get(cat.data_ptr()) if inplace else None
        if (acc is not None and a.cs is not None and b.cs is not None and a.cs[0].data_ptr() == acc.data_ptr() and
                b.cs[0].data_ptr() == acc.data_ptr() and a.cs[1:] == (0, Ca) and b.cs[1:] == (Ca, Cb)):
            out.cs = (acc, 0, Ca + Cb)        # both producers summed into this buffer's accumulator: norm1 has its statistics
        if self.train:
            def bwd():
                self._give(a, out.g[:, :Ca])
                self._give(b, out.g[:, Ca:])
            self.tape.append(bwd)
        return out

    def _skip_view(self, k_, M, C, B=0):
        """(buffer, view) for the producer of skip number k_ (push order): the right C columns of its concat buffer.  The
        buffer gets ONE GroupNorm accumulator for all its columns (both producers add their column sums to it)."""
        ch = self.cat_ch[k_] if k_ < len(self.cat_ch) else None
        if ch is None:
            return None, None
        cat = self._empty(M, ch + C)
        view = cat[:, ch:]
        if self.gn_epi and self._cs_shape_ok(M, B, ch + C) and self.dtype == torch.bfloat16:
            acc = self._cs_alloc(B, ch + C)
            self._cs_cats[cat.data_ptr()] = acc
            self._cs_views[view.data_ptr()] = (acc, ch, cat)
            if ch:
                self._cs_views[cat.data_ptr()] = (acc, 0, cat)
        return cat, view

    @staticmethod
    def _left_view(skips, M, C):
        """View for the producer of the h that the NEXT concat puts in front of the skip on top of the stack, or None."""
        if not skips or skips[-1][1] is None:
            return None
        s, cat = skips[-1]
        if cat.shape[0] != M or cat.shape[1] != C + s.t.shape[1]:
            return None
        return cat[:, :C]

    # ------------------------------------------------------------------ blocks
    def _mark(self, first_key):
        """Tape marker placed at the START of a block: it runs after the whole block's backward, i.e. when every
        gradient from this block's first arena entry to the end of the arena is final (bucketed all-reduce trigger)."""
        if self.train:
            off = self.P.by_key[first_key].off

            def mark():
                self._join_wgrad()         # block boundary: side-stream wgrads of this block are done, operands freed
                if self.grad_ready_cb:     # a consumer that acts on [off, total) calls flush_pending() first
                    self.grad_ready_cb(off)
            self.tape.append(mark)

    def resblock(self, x, r, st, B, H, W, out=None, cs=True):
        """cs: a GroupNorm reads this block's output directly (not through a concat): conv2's epilogue forms its statistics."""
        G = self.cfg.norm_num_groups
        p = r.name
        self._mark(p + ".norm1.weight")
        if self.train and self.group_conv_wgrad:
            self.tape.append(self._wg_flush)
        n1 = self.groupnorm(x, p + ".norm1", B, H * W, G, r.cin // G, 1e-5, True)
        h1, _, _ = self.conv3(n1, p + ".conv1", B, H, W, 0, p + ".conv1.bias", rowvec=st, rv_cols=self.temb_lay[p][:2], cs=True)
        n2 = self.groupnorm(h1, p + ".norm2", B, H * W, r.groups2(G), r.cout // G, 1e-5, True)
        res = x if r.cin == r.cout else self.linear(x, p + ".conv_shortcut", bias=p + ".conv_shortcut.bias")
        y, _, _ = self.conv3(n2, p + ".conv2", B, H, W, 0, p + ".conv2.bias", residual=res, out=out, cs=cs)
        if self.train and self.group_conv_wgrad:
            self.tape.append(self._wg_open)
        return y

    def transformer(self, x, a, ehs, B, H, W, T, out=None, cs=True):
        G = self.cfg.norm_num_groups
        p, c, N = a.name, a.c, H * W
        t = p + ".transformer_blocks.0"
        d1, d2 = a.h1() * 64, a.h2() * 64
        self._mark(p + ".norm.weight")
        if self.train and self.group_wgrad:
            self.tape.append(self._wg_flush)       # runs at the end of the block's backward, before the mark above
        n = self.groupnorm(x, p + ".norm", B, N, G, c // G, 1e-6, False)
        h = self.linear(n, p + ".proj_in", bias=p + ".proj_in.bias")
        qkv = self.linear(h, t + ".attn1.to_qkv", ln=t + ".norm1")
        if self.attn_fp8:            # in place: the backward pass recomputes the scores from the same rounded operands
            k.quantize_e4m3_(qkv.t)
        o = self.attention(qkv.t[:, :d1], qkv.t[:, d1:2 * d1], qkv.t[:, 2 * d1:3 * d1], B, a.h1(), N, N, qkv, qkv,
                           (0, d1), ((d1, 2 * d1), (2 * d1, 3 * d1)), tag=t + ".attn1")
        h = self.linear(o, t + ".attn1.to_out.0", bias=t + ".attn1.to_out.0.bias", residual=h)
        q = self.linear(h, t + ".attn2.to_q", ln=t + ".norm2")
        if self.attn_fp8:
            k.quantize_e4m3_(q.t)
        kv, ko = ehs, self.kv_lay[p][0]      # `ehs` = the batched K/V projection of all transformers; this one's columns
        o = self.attention(q.t[:, :d2], kv.t[:, ko:ko + d2], kv.t[:, ko + d2:ko + 2 * d2], B, a.h2(), N, T, q, kv, (0, d2),
                           ((ko, ko + d2), (ko + d2, ko + 2 * d2)), tag=t + ".attn2")
        h = self.linear(o, t + ".attn2.to_out.0", bias=t + ".attn2.to_out.0.bias", residual=h)
        gl = self.linear(h, t + ".ff.net.0.proj", bias=t + ".ff.net.0.proj.bias", geglu=True, ln=t + ".norm3")
        h = self.linear(gl, t + ".ff.net.2", bias=t + ".ff.net.2.bias", residual=h)
        y = self.linear(h, p + ".proj_out", bias=p + ".proj_out.bias", residual=x, out=out, cs=(B, N) if cs else None)
        if self.train and self.group_wgrad:
            self.tape.append(self._wg_open)        # runs first in the block's backward
        return y

    def _wg_open(self):
        self._wg_items = []

    def _wg_flush(self):
        items, self._wg_items = self._wg_items, None
        if items:
            k.wgrad_group(items, self.slabs)

    # ------------------------------------------------------------------ whole model
    def forward(self, x, timesteps, ehs, B, H, W, train):
        """x: [B*H*W, padc(in_channels)] NHWC rows in self.dtype; timesteps int64 [B]; ehs: [B*T, ctx] in self.dtype.
        Returns (pred Act [B*H*W, padc(out_channels)], acts {d0..,m,u0..: Act})."""
        cfg = self.cfg
        self.train = train
        self.tape = []
        self._cs_begin()
        T = ehs.shape[0] // B
        c0 = cfg.block_out_channels[0]
        k.TAG = "time_embedding"
        te = self._empty(B, c0)
        k.timestep_embed(timesteps, self.freqs, te, B, c0)
        e1 = self.linear(Act(te, rg=False), "time_embedding.linear_1", bias="time_embedding.linear_1.bias")
        temb = self.linear(self.silu(e1), "time_embedding.linear_2", bias="time_embedding.linear_2.bias")
        st = self.silu(temb)                      # shared by every ResBlock (blocks.py:336)
        if self.temb_cols:                        # all time_emb_proj in one skinny GEMM; ResBlocks take column slices
            st = self.linear(st, "time_emb_proj_all", bias="time_emb_proj_all.bias", out_f32=True)
        ehs_act = Act(ehs, rg=False)
        if self.kv_cols:                          # all cross-attention K/V projections in one GEMM; layers take column slices
            ehs_act = self.linear(ehs_act, "attn2_kv_all")
            ehs_act.rg = train
            if self.attn_fp8:        # the K / V of every cross attention, once
                k.quantize_e4m3_(ehs_act.t)
        c0p = padc(c0)
        nskip = 0

        def push(act, cat):         # (skip tensor, its concat buffer or None)
            nonlocal nskip
            skips.append((act, cat))
            nskip += 1

        skips = []
        cat, view = self._skip_view(nskip, B * H * W, c0p, B)
        h, _, _ = self.conv3(Act(x, rg=False), "conv_in", B, H, W, 0, "conv_in.bias", out=view, cs=True)
        push(h, cat)
        acts = {}
        for b in self.blocks:
            cb = padc(b.c)
            if b.kind == "down":
                for j, r in enumerate(b.resnets):
                    att = b.attns[j] if (b.attns and not b.attns[j].dropped) else None
                    cat, view = self._skip_view(nskip, B * H * W, cb, B)
                    made = False
                    if not r.dropped:
                        h = self.resblock(h, r, st, B, H, W, out=None if att is not None else view)
                        made = att is None
                    if att is not None:
                        h = self.transformer(h, att, ehs_act, B, H, W, T, out=view)
                        made = True
                    # both layers dropped: the skip IS the previous tensor, which has no concat buffer for this consumer
                    push(h, cat if made else None)
                if b.sampler:
                    cat, view = self._skip_view(nskip, B * ((H + 1) // 2) * ((W + 1) // 2), cb, B)
                    h, H, W = self.conv3(h, f"{b.name}.downsamplers.0.conv", B, H, W, 1,
                                         f"{b.name}.downsamplers.0.conv.bias", out=view, cs=True)
                    push(h, cat)
                acts[f"d{b.idx}"] = h
            elif b.kind == "mid":
                h = self.resblock(h, b.resnets[0], st, B, H, W)
                h = self.transformer(h, b.attns[0], ehs_act, B, H, W, T)
                h = self.resblock(h, b.resnets[1], st, B, H, W, out=self._left_view(skips, B * H * W, cb))
                acts["m"] = h
            else:
                n = len(b.resnets)
                for j, r in enumerate(b.resnets):
                    s, scat = skips.pop()
                    att = b.attns[j] if (b.attns and not b.attns[j].dropped) else None
                    # the tensor this pair leaves behind is the left half of the next concat (unless an upsampler follows)
                    nxt = None if (j == n - 1 and b.sampler) else self._left_view(skips, B * H * W, cb)
                    # (ahead of an upsampler no GroupNorm reads the tensor: no statistics for it)
                    last = j == n - 1 and b.sampler
                    if not r.dropped:      # dropped: keep the non-skip channels == h itself (blocks.py:502-515)
                        h = self.resblock(self.concat(h, s, scat), r, st, B, H, W, out=None if att is not None else nxt,
                                          cs=att is not None or not last)
                    if att is not None:
                        h = self.transformer(h, att, ehs_act, B, H, W, T, out=nxt, cs=not last)
                if b.sampler:
                    h, H, W = self.conv3(h, f"{b.name}.upsamplers.0.conv", B, H, W, 2,
                                         f"{b.name}.upsamplers.0.conv.bias",
                                         out=self._left_view(skips, B * 4 * H * W, cb), cs=True)
                acts[f"u{b.idx}"] = h
        assert not skips
        n = self.groupnorm(h, "conv_norm_out", B, H * W, cfg.norm_num_groups, c0 // cfg.norm_num_groups, 1e-5, True)
        pred, _, _ = self.conv3(n, "conv_out", B, H, W, 0, "conv_out.bias")
        return pred, acts

    def backward(self):
        """Replays the tape; the caller has seeded .g of the loss inputs (pred and, optionally, block activations)."""
        tape, self.tape = self.tape, []
        for fn in reversed(tape):
            fn()
        self._join_wgrad()
        self.flush_pending()
