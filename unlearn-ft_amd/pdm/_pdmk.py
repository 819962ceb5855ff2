"""ctypes binding of libpdmk.so (include/pdmk.h) — the only route from the Python host side to the HIP kernels.

There is NO fallback: if the shared library is missing or a call returns a non-zero status this module raises.
torch is used for device memory and streams only (tensors are passed as raw device pointers).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PDMK_LIB: another build of the same library (same-box A/B of kernel variants, tools/); never a fallback - it must exist
LIB_PATH = os.environ.get("PDMK_LIB") or os.path.join(os.path.dirname(_HERE), "libpdmk.so")

F32, BF16 = 0, 1
EPI_NONE, EPI_GEGLU, EPI_GEGLU_BWD = 0, 1, 2
A_ROWK, A_CONV, A_COLK = 0, 1, 2
B_ROWK, B_COLK, B_COLK_CONV = 0, 1, 2


class PdmkError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise PdmkError(f"libpdmk.so not found at {LIB_PATH}: build it with `python __graft_entry__.py` "
                        f"(or `make -C unlearn-ft_amd/csrc`); there is no CPU/PyTorch fallback for the hot path")
    return C.CDLL(LIB_PATH)


_lib = _load()

vp, i32, i64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double


class GemmArgs(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("bias", vp), ("rowvec", vp), ("R", vp), ("colsum_out", vp),
                ("M", i32), ("N", i32), ("K", i32),
                ("lda", i32), ("ldb", i32), ("ldc", i32), ("ldr", i32),
                ("rows_per_b", i32), ("a_mode", i32), ("b_mode", i32),
                ("conv_b", i32), ("conv_hi", i32), ("conv_wi", i32), ("conv_ci", i32), ("conv_ho", i32),
                ("conv_wo", i32), ("conv_mode", i32), ("conv_ld", i32),
                ("dtype", i32), ("out_f32", i32), ("accumulate", i32), ("splitk", i32), ("alpha", f32), ("ldrv", i32),
                ("epilogue", i32), ("ldc2", i32), ("C2", vp), ("colstat", vp), ("cs_ld", i32), ("cs_col0", i32),
                ("ln_gamma", vp), ("ln_beta", vp), ("ln_stats", vp), ("ln_out", vp), ("ld_ln_out", i32), ("ln_eps", f32)]


_SIGS = {
    "pdmk_version": ([], i32),
    "pdmk_gemm": ([C.POINTER(GemmArgs), vp], i32),
    "pdmk_gemm_group": ([C.POINTER(GemmArgs), i32, vp, C.POINTER(i32)], i32),
    "pdmk_gemm_plan": ([C.POINTER(GemmArgs), vp, C.POINTER(i32)], i32),
    "pdmk_gemm_ln_supported": ([C.POINTER(GemmArgs)], i32),
    "pdmk_conv_up2_supported": ([i32, i32, i32, i32, i32, i32], i32),
    "pdmk_up2_pack_weights": ([vp, vp, vp, i32, i32, i32, vp], i32),
    "pdmk_up2_combine_wgrad": ([vp, vp, i32, i32, vp], i32),
    "pdmk_gemm_last_candidate": ([], i32),
    "pdmk_gemm_candidate_name": ([i32, i32, i32, C.c_char_p, i32], i32),
    "pdmk_splitk_finish": ([vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    "pdmk_splitk_finish_colstat": ([vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp, i32, i32, i32, vp], i32),
    "pdmk_groupnorm_fwd": ([vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, i32, i32, vp], i32),
    "pdmk_groupnorm_apply_colstat": ([vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, i32, i32, vp], i32),
    "pdmk_groupnorm_bwd": ([vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp], i32),
    "pdmk_groupnorm_bwd_partial_dims": ([i32, i32, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32)], i32),
    "pdmk_layernorm_bwd_partial_dims": ([i32, i32, C.POINTER(i32), C.POINTER(i32)], i32),
    "pdmk_reduce_partials_group": ([vp, i32, vp], i32),
    "pdmk_splitk_finish_group": ([vp, i32, vp], i32),
    "pdmk_layernorm_fwd": ([vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp], i32),
    "pdmk_layernorm_bwd": ([vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp], i32),
    "pdmk_attn_fwd": ([vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i32, i64, i32, i64, i32, i64, i32, f32, i32, vp], i32),
    "pdmk_attn_bwd": ([vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32,
                       i64, i32, i64, i32, i64, i32, i64, i32, i64, i32, i64, i32, i64, i32, f32, vp, i64, i32, vp], i32),
    "pdmk_geglu_fwd": ([vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    "pdmk_geglu_bwd": ([vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    "pdmk_quantize_e4m3": ([vp, vp, i64, i32, vp], i32),
    "pdmk_silu_fwd": ([vp, vp, i64, i32, vp], i32),
    "pdmk_silu_bwd": ([vp, vp, vp, i64, i32, vp], i32),
    "pdmk_copy2d": ([vp, vp, i64, i32, i32, i32, i32, i32, vp], i32),
    "pdmk_cast_permute": ([vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "pdmk_colsum": ([vp, vp, i64, i32, i32, i32, i32, i32, i32, vp], i32),
    "pdmk_pool2x2_sum": ([vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "pdmk_timestep_embed": ([vp, vp, vp, i32, i32, i32, vp], i32),
    "pdmk_add_noise_velocity": ([vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "pdmk_nchw_to_nhwc": ([vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "pdmk_nhwc_to_nchw": ([vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "pdmk_mse_fwd": ([vp, i32, vp, i32, vp, vp, i32, i32, i64, i32, i32, i32, f64, vp], i32),
    "pdmk_mse_bwd": ([vp, i32, vp, i32, vp, vp, i32, i64, i32, i32, i32, i32, f32, i32, vp], i32),
    "pdmk_mse_fwd_bwd": ([vp, i32, vp, i32, vp, vp, i32, vp, i32, i64, i32, i32, i32, i32, f64, f32, i32, vp], i32),
    "pdmk_axpby": ([vp, vp, f32, f32, i64, i32, vp], i32),
    "pdmk_adamw": ([vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, vp, f32, i32, vp, vp], i32),
    "pdmk_transpose_tiles": ([vp, vp, vp, i32, i32, vp], i32),
    "pdmk_sumsq": ([vp, i64, vp, i32, vp], i32),
    "pdmk_zero": ([vp, i64, vp], i32),
    "pdmk_skinny_gemm": ([vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    "pdmk_skinny_wgrad": ([vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    "pdmk_softmax_rows": ([vp, vp, i64, i32, i64, i64, i32, vp], i32),
    "pdmk_latent_sample": ([vp, i32, vp, vp, i32, i32, i32, f32, i32, vp], i32),
    "pdmk_embed_tokens": ([vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    "pdmk_attn_fwd_causal": ([vp, vp, vp, vp, vp, i32, i32, i32, i64, i32, i64, i32, i64, i32, i64, i32, f32, i32, vp], i32),
    "pdmk_gelu_fwd": ([vp, vp, i64, i32, vp], i32),
    "pdmk_gemm_splitk_workspace_bytes": ([i64, i32, i32], i64),
    "pdmk_groupnorm_workspace_bytes": ([i32, i32], i64),
    "pdmk_groupnorm_bwd_part_workspace_bytes": ([i32, i32], i64),
    "pdmk_layernorm_bwd_part_workspace_bytes": ([i32, i32], i64),
    "pdmk_attn_bwd_workspace_bytes": ([i32, i32, i32, i32], i64),
    "pdmk_plan_export": ([C.c_char_p], i32),
    "pdmk_plan_import": ([C.c_char_p], i32),
    "pdmk_plan_size": ([], i32),
    "pdmk_plan_clear": ([], i32),
    "pdmk_debug_scratch_violations": ([], i32),
    "pdmk_comm_unique_id": ([vp], i32),
    "pdmk_comm_create": ([vp, i32, i32, C.POINTER(vp)], i32),
    "pdmk_comm_allreduce_sum_f32": ([vp, vp, i64, vp], i32),
    "pdmk_comm_world": ([vp], i32),
    "pdmk_comm_rank": ([vp], i32),
    "pdmk_comm_reduce_scatter_sum_f32": ([vp, vp, i64, vp], i32),
    "pdmk_comm_allgather_f32": ([vp, vp, i64, vp], i32),
    "pdmk_stream_create": ([i32, C.POINTER(vp)], i32),
    "pdmk_stream_destroy": ([vp], i32),
    "pdmk_comm_destroy": ([vp], i32),
}
for _n, (_a, _r) in _SIGS.items():
    _f = getattr(_lib, _n)          # AttributeError here = header/library mismatch: fail at import
    _f.argtypes, _f.restype = _a, _r

EXPORTS = tuple(_SIGS)


def version():
    return _lib.pdmk_version()


def dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise PdmkError(f"unsupported dtype {t.dtype}")


def _p(t):
    return None if t is None else t.data_ptr()


def _st():
    return torch.cuda.current_stream().cuda_stream


def _chk(rc, name):
    if rc != 0:
        raise PdmkError(f"{name} failed with status {rc}")


PROFILE = None   # bench.py sets this to a list: every gemm launch is then bracketed by HIP events on the launch stream

# ---- lockstep recording (teacher || student forward, pdmk_gemm_group): while RECORD is a list every launch wrapper below
# appends a record instead of launching; run_lockstep() then walks two record lists side by side and issues the launches of
# both, pairing what the library can serve in one grouped launch.  Host-side work (allocation, planning queries) still
# happens at record time; records keep their operand tensors alive until they have run.
RECORD = None
TAG = ""         # set by the engine before every layer op: records of the same layer op of two models share a tag
GROUP_MAX = 8
STATS = {"launches": 0, "grouped": 0}      # launches issued through records since the last reset (tests / bench bookkeeping)


class Rec:
    __slots__ = ("tag", "kind", "fn", "g", "keep", "macs", "meta")

    def __init__(self, kind, fn, g=None, keep=(), macs=None, meta=None):
        self.tag, self.kind, self.fn, self.g, self.keep, self.macs, self.meta = TAG, kind, fn, g, keep, macs, meta

    def run(self):
        self.fn()


class Recorder:
    """with Recorder() as r: ...launch wrappers record...;  r.recs is the list."""

    def __enter__(self):
        global RECORD
        self.prev, self.recs = RECORD, []
        RECORD = self.recs
        return self

    def __exit__(self, *exc):
        global RECORD
        RECORD = self.prev
        return False


def _recordable(kind):
    """Launch wrapper that defers itself while recording (generic: replayed as a closure, never grouped)."""
    def deco(fn):
        def wrapped(*a, **kw):
            if RECORD is not None:
                RECORD.append(Rec(kind, lambda: fn(*a, **kw)))
                return None
            return fn(*a, **kw)
        wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
        return wrapped
    return deco


def _launch_gemm(g, macs, shape):
    """pdmk_gemm on a filled argument block (+ the optional HIP-event bracket of bench.py's per-kernel profile)."""
    if PROFILE is None:
        _chk(_lib.pdmk_gemm(C.byref(g), _st()), "pdmk_gemm")
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _chk(_lib.pdmk_gemm(C.byref(g), _st()), "pdmk_gemm")
    e1.record()
    kind = ("bf16" if g.dtype == BF16 else "f32", g.a_mode, g.b_mode, _lib.pdmk_gemm_last_candidate())
    PROFILE.append((kind, 2.0 * (macs if macs is not None else g.M * g.N * g.K), e0, e1, shape))


def gemm_group(recs):
    """Issue the GEMM records `recs` (2..GROUP_MAX independent problems) through pdmk_gemm_group: one launch where the
    library has a kernel shape for all of them (bit-identical to the separate launches), else one by one."""
    n = len(recs)
    arr = (GemmArgs * n)()
    for i, r in enumerate(recs):
        arr[i] = r.g
    got = i32(0)
    prof = PROFILE is not None
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _chk(_lib.pdmk_gemm_group(arr, n, _st(), C.byref(got)), "pdmk_gemm_group")
    STATS["launches"] += 1 if got.value == n else n
    STATS["grouped"] += got.value
    if prof:
        e1.record()
        g0 = recs[0].g
        kind = ("bf16" if g0.dtype == BF16 else "f32", g0.a_mode, g0.b_mode, _lib.pdmk_gemm_last_candidate(), int(got.value))
        flops = sum(2.0 * (r.macs if r.macs is not None else r.g.M * r.g.N * r.g.K) for r in recs)
        PROFILE.append((kind, flops, e0, e1, [(r.g.M, r.g.N, r.g.K, int(r.g.splitk), int(bool(r.g.R) or r.g.accumulate == 1))
                                               for r in recs]))
    return int(got.value)


def run_lockstep(ta, tb, group=True):
    """Issue two recorded launch sequences (each in its own order) side by side: records of both lists with the same tag and
    kind go out together - as ONE launch when they are GEMMs the library groups - everything else one by one.  Any
    interleaving that keeps each list's order is valid (the two sequences are independent)."""
    last = {}
    for idx, r in enumerate(ta):
        last[(r.tag, r.kind)] = idx
    i = j = 0
    while i < len(ta) or j < len(tb):
        if i < len(ta) and j < len(tb) and ta[i].tag == tb[j].tag and ta[i].kind == tb[j].kind:
            a, b = ta[i], tb[j]
            i, j = i + 1, j + 1
            if group and a.kind == "gemm":
                gemm_group([a, b])
            elif group and a.kind in _GROUP_FNS:
                _GROUP_FNS[a.kind]([a, b])
                STATS["launches"] += 1
                STATS["grouped"] += 2
            else:
                a.run()
                b.run()
                STATS["launches"] += 2
        elif j < len(tb) and (i >= len(ta) or last.get((tb[j].tag, tb[j].kind), -1) < i):
            tb[j].run()
            j += 1
            STATS["launches"] += 1
        else:
            ta[i].run()
            i += 1
            STATS["launches"] += 1


_GROUP_FNS = {}     # kind -> fn(list of Rec): grouped forms of the non-GEMM forward kernels (filled in below)


def gemm(A, B, Cout, M, N, K, lda, ldb, ldc, *, bias=None, rowvec=None, rows_per_b=0, R=None, ldr=0,
         a_mode=A_ROWK, b_mode=B_ROWK, conv=None, dtype=None, out_f32=False, accumulate=False, splitk=1, alpha=1.0,
         macs=None, colsum_out=None, ldrv=0, epilogue=EPI_NONE, C2=None, ldc2=0, colstat=None, ln=None):
    # accumulate: False / True / 2 (= split-K slabs, see pdmk.h)
    # ln = (gamma, beta, stats or None, out or None, eps): LayerNorm(A) in the GEMM's prologue (pdmk_gemm_args.ln_gamma)
    """conv = (b, hi, wi, ci, ho, wo, mode, ld) or None.  macs: logical (un-padded) multiply-accumulates, profiling only."""
    g = GemmArgs()
    g.colsum_out = _p(colsum_out)
    g.A, g.B, g.C = _p(A), _p(B), _p(Cout)
    g.bias, g.rowvec, g.R = _p(bias), _p(rowvec), _p(R)
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc, g.ldr = lda, ldb, ldc, ldr
    g.rows_per_b, g.ldrv = rows_per_b, ldrv
    g.a_mode, g.b_mode = a_mode, b_mode
    if conv is not None:
        (g.conv_b, g.conv_hi, g.conv_wi, g.conv_ci, g.conv_ho, g.conv_wo, g.conv_mode, g.conv_ld) = conv
    g.dtype = dt(A) if dtype is None else dtype
    g.out_f32, g.accumulate, g.splitk, g.alpha = int(out_f32), int(accumulate), int(splitk), float(alpha)
    g.epilogue, g.C2, g.ldc2 = int(epilogue), _p(C2), int(ldc2)
    if colstat is not None:          # (accumulator [B, 4, cs_ld] int64 limbs, first accumulator column of this output)
        g.colstat, g.cs_ld, g.cs_col0 = _p(colstat[0]), colstat[0].shape[2], int(colstat[1])
    _set_ln(g, ln)
    shape = (M, N, K, int(splitk), int(R is not None or int(accumulate) == 1))     # last: the epilogue also READS an [M, N] tensor
    if RECORD is not None:
        RECORD.append(Rec("gemm", lambda: _launch_gemm(g, macs, shape), g=g, macs=macs,
                          keep=(A, B, Cout, bias, rowvec, R, colsum_out, C2, colstat, ln)))
        return
    _launch_gemm(g, macs, shape)


def _set_ln(g, ln):
    if ln is None:
        return
    gamma, beta, stats, out, eps = ln
    g.ln_gamma, g.ln_beta, g.ln_stats, g.ln_out = _p(gamma), _p(beta), _p(stats), _p(out)
    g.ld_ln_out, g.ln_eps = (0 if out is None else out.stride(0)), float(eps)


def gemm_ln_supported(A, B, M, N, K, lda, ldb, *, geglu=False, residual=False, bias=False):
    """Would pdmk_gemm take this Linear with the LayerNorm of its input fused into the prologue (one launch)?  Shape question only:
    the pointers are placeholders."""
    g = GemmArgs()
    g.A, g.B, g.C = _p(A), _p(B), _p(A)
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc = M, N, K, lda, ldb, (N // 2 if geglu else N)
    g.a_mode, g.b_mode, g.dtype, g.splitk, g.alpha = A_ROWK, B_ROWK, dt(A), 1, 1.0
    g.epilogue = EPI_GEGLU if geglu else EPI_NONE
    if residual:
        g.R, g.ldr = _p(A), N
    if bias or geglu:
        g.bias = _p(B)
    g.ln_gamma = g.ln_beta = _p(B)
    g.ln_eps = 1e-5
    return bool(_lib.pdmk_gemm_ln_supported(C.byref(g)))


def _ws_bytes(n):
    if n < 0:
        raise PdmkError("workspace query rejected its arguments")
    return int(n)


def plan_export(path):
    _chk(_lib.pdmk_plan_export(os.fsencode(path)), "pdmk_plan_export")


def plan_import(path):
    n = _lib.pdmk_plan_import(os.fsencode(path))
    if n < 0:
        raise PdmkError(f"pdmk_plan_import({path!r}) failed with status {n}")
    return n


def plan_size():
    return int(_lib.pdmk_plan_size())


def plan_clear():
    _chk(_lib.pdmk_plan_clear(), "pdmk_plan_clear")


def debug_scratch_violations():
    """Tuning passes that overran their scratch since the process started (PDMK_DEBUG_SCRATCH=1; else 0)."""
    return int(_lib.pdmk_debug_scratch_violations())


class Comm:
    """pdmk_comm_t: RCCL communicator behind the C ABI (one per rank; `uid` = the 128 bytes rank 0 got from unique_id())."""

    def __init__(self, uid, rank, world):
        h = vp()
        buf = C.create_string_buffer(bytes(uid), 128)
        _chk(_lib.pdmk_comm_create(buf, rank, world, C.byref(h)), "pdmk_comm_create")
        self._h, self.rank, self.world = h, rank, world

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _chk(_lib.pdmk_comm_unique_id(buf), "pdmk_comm_unique_id")
        return bytes(buf.raw)

    def world_size(self):
        """What the HANDLE says (pdmk_comm_world / pdmk_comm_rank), not what the constructor was told."""
        return int(_lib.pdmk_comm_world(self._h))

    def rank_id(self):
        return int(_lib.pdmk_comm_rank(self._h))

    def all_reduce_sum_(self, t):
        """In-place sum over the ranks of a contiguous fp32 tensor, asynchronous on torch's current HIP stream."""
        assert t.dtype == torch.float32 and t.is_contiguous()
        _chk(_lib.pdmk_comm_allreduce_sum_f32(self._h, _p(t), t.numel(), _st()), "pdmk_comm_allreduce_sum_f32")

    def reduce_scatter_sum_(self, t):
        """t: contiguous fp32 bucket of world * n elements; afterwards this rank's share t[rank*n:(rank+1)*n] holds the sum."""
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() % self.world == 0
        _chk(_lib.pdmk_comm_reduce_scatter_sum_f32(self._h, _p(t), t.numel() // self.world, _st()),
             "pdmk_comm_reduce_scatter_sum_f32")

    def all_gather_(self, t):
        """Completes every rank's share of the bucket on every rank (second half of the all-reduce)."""
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() % self.world == 0
        _chk(_lib.pdmk_comm_allgather_f32(self._h, _p(t), t.numel() // self.world, _st()), "pdmk_comm_allgather_f32")

    def close(self):
        if self._h is not None:
            _lib.pdmk_comm_destroy(self._h)
            self._h = None


_ROLE_STREAMS = {}


def role_stream(device, role, high_priority=False):
    """The process-wide dedicated HIP stream of a role ("teacher", "opt", "wt", "comm", "capture", ...) on `device`:
    created once through pdmk_stream_create and wrapped for torch - never one of torch's 32 pooled streams, which are handed
    out round-robin and start to alias after a few stepper / graph instances.  Never destroyed (work may be queued on it)."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), role)
    s = _ROLE_STREAMS.get(key)
    if s is None:
        h = vp()
        with torch.cuda.device(key[0]):
            _chk(_lib.pdmk_stream_create(int(high_priority), C.byref(h)), "pdmk_stream_create")
        s = torch.cuda.ExternalStream(h.value, device=torch.device("cuda", key[0]))
        _ROLE_STREAMS[key] = s
    return s


def groupnorm_ws(device, B, G, have=None):
    """fp64 scratch of pdmk_groupnorm_fwd / _bwd, sized by the library; `have` is returned when it is already big enough."""
    need = _ws_bytes(_lib.pdmk_groupnorm_workspace_bytes(B, G)) // 8
    if have is not None and have.numel() >= need:
        return have
    return torch.empty(max(need, 1 << 17), device=device, dtype=torch.float64)


def zeros(shape, device, dtype):
    """torch.zeros without a memset node (see pdmk_zero in include/pdmk.h); byte size padded to 16."""
    n = 1
    for d in shape:
        n *= d
    esz = torch.empty((), dtype=dtype).element_size()
    pad = (-(n * esz)) % 16 // esz
    flat = torch.empty(n + pad, device=device, dtype=dtype)
    _chk(_lib.pdmk_zero(_p(flat), (n + pad) * esz, _st()), "pdmk_zero")
    return flat[:n].view(*shape)


def zero_(t):
    assert t.is_contiguous() and (t.numel() * t.element_size()) % 16 == 0
    _chk(_lib.pdmk_zero(_p(t), t.numel() * t.element_size(), _st()), "pdmk_zero")
    return t


def last_candidate():
    """Candidate id of the calling thread's last pdmk_gemm launch (0 = K-step-32 kernels, 1.. = ring / halo shapes)."""
    return int(_lib.pdmk_gemm_last_candidate())


def candidate_name(a_mode, b_mode, cand):
    buf = C.create_string_buffer(160)
    _chk(_lib.pdmk_gemm_candidate_name(a_mode, b_mode, cand, buf, 160), "pdmk_gemm_candidate_name")
    return buf.value.decode()


def splitk_plan(A, B, M, N, K, lda, ldb, a_mode=A_ROWK, conv=None):
    """Split-K factor for a forward/dgrad GEMM, from the library's plan cache (tuned on first sight of the shape)."""
    g = GemmArgs()
    g.A, g.B = _p(A), _p(B)
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = lda, ldb, N
    g.a_mode, g.b_mode = a_mode, B_ROWK
    if conv is not None:
        (g.conv_b, g.conv_hi, g.conv_wi, g.conv_ci, g.conv_ho, g.conv_wo, g.conv_mode, g.conv_ld) = conv
    g.dtype = dt(A)
    g.splitk, g.alpha = 1, 1.0
    sk = i32(1)
    _chk(_lib.pdmk_gemm_plan(C.byref(g), _st(), C.byref(sk)), "pdmk_gemm_plan")
    return int(sk.value)


def wgrad_plan(dy, x, M, N, K, lda, ldb, b_mode=B_COLK, conv=None, slabs=False):
    """Split-K factor for a weight-gradient GEMM dW[M,N] += dy[K,M]^T x[K,N] (both operands reduction-major).
    slabs: the splits will store partial slabs (accumulate = 2) instead of adding with atomics - planned apart."""
    g = GemmArgs()
    g.accumulate = 2 if slabs else 0
    g.A, g.B = _p(dy), _p(x)
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = lda, ldb, N
    g.a_mode, g.b_mode = A_COLK, b_mode
    if conv is not None:
        (g.conv_b, g.conv_hi, g.conv_wi, g.conv_ci, g.conv_ho, g.conv_wo, g.conv_mode, g.conv_ld) = conv
    g.dtype = dt(x)
    g.out_f32, g.splitk, g.alpha = 1, 1, 1.0
    sk = i32(1)
    _chk(_lib.pdmk_gemm_plan(C.byref(g), _st(), C.byref(sk)), "pdmk_gemm_plan")
    return int(sk.value)


def gemm_auto(A, B, Cout, M, N, K, lda, ldb, ldc, *, bias=None, rowvec=None, rows_per_b=0, R=None, ldr=0,
              a_mode=A_ROWK, conv=None, accumulate=False, macs=None, ldrv=0, colstat=None):
    """Forward / dgrad GEMM with the split-K decision made by the planner: split shapes go through an fp32 workspace.
    colstat: (accumulator, first column): GroupNorm statistics of the stored output from the GEMM's epilogue, or from the finish
    pass of a split plan (M % 64 == 0 and rows_per_b % 64 == 0, bf16).  Returns True when the statistics were accumulated."""
    sk = splitk_plan(A, B, M, N, K, lda, ldb, a_mode, conv)
    if sk == 1:
        gemm(A, B, Cout, M, N, K, lda, ldb, ldc, bias=bias, rowvec=rowvec, rows_per_b=rows_per_b, R=R, ldr=ldr,
             a_mode=a_mode, conv=conv, accumulate=accumulate, macs=macs, ldrv=ldrv, colstat=colstat)
        return colstat is not None
    # split-K: every split stores its fp32 partial into its own slab (plain stores: no atomics, no zero-fill, the sum
    # order is fixed), the finish pass adds the slabs and applies the epilogue
    ws = torch.empty(_ws_bytes(_lib.pdmk_gemm_splitk_workspace_bytes(M, N, sk)) // 4, device=A.device, dtype=torch.float32)
    gemm(A, B, ws, M, N, K, lda, ldb, N, a_mode=a_mode, conv=conv, out_f32=True, splitk=sk, accumulate=2, macs=macs)
    splitk_finish(ws, Cout, M, N, ldc, sk, bias=bias, rowvec=rowvec, R=R, ldr=ldr, rows_per_b=rows_per_b, ldrv=ldrv,
                  accumulate=accumulate, colstat=colstat)
    return colstat is not None


@_recordable("splitk_finish")
def splitk_finish(ws, Cout, M, N, ldc, nslab, *, bias=None, rowvec=None, R=None, ldr=0, rows_per_b=0, ldrv=0,
                  accumulate=False, colstat=None):
    """colstat = (accumulator [B, 4, ld] int64, first column): the GroupNorm statistics of the stored output leave with this pass."""
    if colstat is not None:
        _chk(_lib.pdmk_splitk_finish_colstat(_p(ws), _p(Cout), _p(bias), _p(rowvec), _p(R), M, N, ldc, ldr, rows_per_b, ldrv,
                                             nslab, int(accumulate), _p(colstat[0]), colstat[0].shape[2], int(colstat[1]),
                                             dt(Cout), _st()), "pdmk_splitk_finish_colstat")
        return
    _chk(_lib.pdmk_splitk_finish(_p(ws), _p(Cout), _p(bias), _p(rowvec), _p(R), M, N, ldc, ldr, rows_per_b, ldrv, nslab,
                                 int(accumulate), dt(Cout), _st()), "pdmk_splitk_finish")


class SlabItem(C.Structure):
    _fields_ = [("ws", vp), ("dst", vp), ("n", i64), ("nslab", i32), ("pad_", i32)]


SLAB_GROUP_MAX = 32


class SlabQueue:
    """Deferred sums of weight-gradient split-K slabs (pdmk_splitk_finish_group): a weight gradient whose splits stored
    their partials into a [sk][M][N] workspace is added into its gradient later, up to 32 weights per launch."""

    def __init__(self, max_bytes=1 << 30):
        self.items, self.bytes, self.max_bytes = [], 0, max_bytes

    def add(self, ws, dW, n, nslab):
        self.items.append((ws, dW, n, nslab))       # ws stays referenced until the flush
        self.bytes += ws.numel() * 4

    def full(self):
        return len(self.items) >= SLAB_GROUP_MAX or self.bytes >= self.max_bytes

    def flush(self):
        while self.items:
            chunk, self.items = self.items[:SLAB_GROUP_MAX], self.items[SLAB_GROUP_MAX:]
            arr = (SlabItem * len(chunk))()
            for a, (ws, dW, n, nslab) in zip(arr, chunk):
                a.ws, a.dst, a.n, a.nslab = _p(ws), _p(dW), n, nslab
            _chk(_lib.pdmk_splitk_finish_group(C.cast(arr, vp), len(chunk), _st()), "pdmk_splitk_finish_group")
        self.bytes = 0


def wgrad(dy, x, dW, M, N, K, lda, ldb, *, b_mode=B_COLK, conv=None, colsum_out=None, macs=None, queue=None):
    """dW[M, N] (fp32, row stride N) += dy[K, M]^T x[K, N] (3x3 gather of x for b_mode = B_COLK_CONV), split over the
    pixel dimension K as the planner says.  Without a queue the splits add into dW with fp32 atomics (a slab + finish pass
    PER WEIGHT was measured 4 % slower for the step: one more launch per weight outweighs the atomics).  queue (a
    SlabQueue): the splits store partial slabs with plain stores and the queue adds them into dW at its next flush, many
    weights per launch - no atomics and no per-weight launch."""
    if queue is not None and (M * N) % 4 == 0 and dW.is_contiguous():
        sk = wgrad_plan(dy, x, M, N, K, lda, ldb, b_mode, conv, slabs=True)
        if sk > 1:
            if queue.full():
                queue.flush()
            ws = torch.empty(sk * M * N, device=dy.device, dtype=torch.float32)
            gemm(dy, x, ws, M, N, K, lda, ldb, N, a_mode=A_COLK, b_mode=b_mode, conv=conv, out_f32=True, splitk=sk,
                 accumulate=2, dtype=dt(x), macs=macs, colsum_out=colsum_out)
            queue.add(ws, dW, M * N, sk)
            return
    else:
        sk = wgrad_plan(dy, x, M, N, K, lda, ldb, b_mode, conv)
    gemm(dy, x, dW, M, N, K, lda, ldb, N, a_mode=A_COLK, b_mode=b_mode, conv=conv, out_f32=True, splitk=sk,
         accumulate=(sk == 1), dtype=dt(x), macs=macs, colsum_out=colsum_out)


_WG_TARGET = int(os.environ.get("PDMK_WG_TARGET", "512"))     # workgroups a block's grouped weight-gradient launch aims for
_WG_MINK = int(os.environ.get("PDMK_WG_MINK", "32"))           # and the fewest 64-row K-steps a split may be left with
_WG_CONV_SLABS = os.environ.get("PDMK_WGRAD_SLABS_CONV", "0") == "1"   # grouped 3x3 conv weight gradients: split partials to slabs (1) or fp32 atomics (0)


def wgrad_group(items, queue, target_wgs=None):
    """The weight gradients of one block (blocks.py:705-867 backward: to_q/k/v, to_out, ff.net.0.proj, ff.net.2, proj_in, proj_out
    of a transformer block; blocks.py:308-381 backward: conv1 / conv2 of a ResBlock; reached from accelerator.backward,
    trainer.py:2782) as grouped launches (pdmk_gemm_group): every item reduces over the SAME K pixel rows into a small [M, N]
    output, so one problem alone fills the 256 CUs only by cutting its reduction into 16-32 splits (16 K-steps each behind a cold
    prologue, 16-32 slabs to add); together the problems have the tiles, so they share ONE split factor chosen for the group
    (>= _WG_MINK K-steps per split) and one launch.
    items: (dy, x, dW, M, N, K, lda, ldb, colsum_out, macs[, b_mode, conv]) with dW fp32 [M, N] contiguous; Linear items
    (b_mode B_COLK) store the slabs of a split reduction for `queue` (a SlabQueue) to add later, 3x3 conv items (B_COLK_CONV) add
    their splits with atomics as k.wgrad does.  Problems the grouped kernels do not take go out one by one (same results)."""
    items = [tuple(it) + (B_COLK, None) if len(it) == 10 else tuple(it) for it in items]
    while items:
        K, bm = items[0][5], items[0][10]
        same = [it for it in items if it[5] == K and it[10] == bm][:GROUP_MAX]
        items = [it for it in items if not any(it is s_ for s_ in same)]
        conv_g = bm == B_COLK_CONV
        tiles = sum(((it[3] + 127) // 128) * ((it[4] + 127) // 128) for it in same)
        nk = max(1, K // 64)
        sk = max(1, min((target_wgs or _WG_TARGET) // max(tiles, 1), nk // _WG_MINK, 64))
        if len(same) == 1 or (queue is None and not conv_g):
            for dy, x, dW, M, N, K_, lda, ldb, cs, macs, bm_, conv in same:
                wgrad(dy, x, dW, M, N, K_, lda, ldb, b_mode=bm_, conv=conv, colsum_out=cs, macs=macs, queue=None if conv_g else queue)
            continue
        with Recorder() as r:
            slabs = []
            for dy, x, dW, M, N, K_, lda, ldb, cs, macs, bm_, conv in same:
                if (not conv_g or _WG_CONV_SLABS) and queue is not None and sk > 1 and (M * N) % 4 == 0 and dW.is_contiguous():
                    ws = torch.empty(sk * M * N, device=dy.device, dtype=torch.float32)
                    gemm(dy, x, ws, M, N, K_, lda, ldb, N, a_mode=A_COLK, b_mode=bm_, conv=conv, out_f32=True, splitk=sk, accumulate=2,
                         dtype=dt(x), macs=macs, colsum_out=cs)
                    slabs.append((ws, dW, M * N, sk))
                else:       # unsplit: added into the gradient in the epilogue; conv splits: fp32 atomics into it
                    ska = sk if conv_g else 1
                    gemm(dy, x, dW, M, N, K_, lda, ldb, N, a_mode=A_COLK, b_mode=bm_, conv=conv, out_f32=True, splitk=ska,
                         accumulate=(ska == 1), dtype=dt(x), macs=macs, colsum_out=cs)
        gemm_group(r.recs)
        for ws, dW, n, nslab in slabs:
            if queue.full():
                queue.flush()
            queue.add(ws, dW, n, nslab)


@_recordable("groupnorm_apply_colstat")
def groupnorm_apply_colstat(x, y, gamma, beta, stats, colstat, col0, B, HW, Cc, ldx, ldy, G, gs, eps, silu):
    """GroupNorm(+SiLU) forward with the statistics taken from a producing GEMM's epilogue sums (colstat [B, 4, cs_ld] int64 limbs)."""
    _chk(_lib.pdmk_groupnorm_apply_colstat(_p(x), _p(y), _p(gamma), _p(beta), _p(stats), _p(colstat), colstat.shape[2], int(col0),
                                           B, HW, Cc, ldx, ldy, G, gs, eps, int(silu), dt(x), _st()),
         "pdmk_groupnorm_apply_colstat")


@_recordable("groupnorm_fwd")
def groupnorm_fwd(x, y, gamma, beta, stats, ws, B, HW, Cc, ldx, ldy, G, gs, eps, silu):
    _chk(_lib.pdmk_groupnorm_fwd(_p(x), _p(y), _p(gamma), _p(beta), _p(stats), _p(ws), B, HW, Cc, ldx, ldy, G, gs,
                                 eps, int(silu), dt(x), _st()), "pdmk_groupnorm_fwd")


_PART_WS = {}


def part_ws(device, elems):
    """Per-device fp32 scratch for the two-stage per-channel gradient reductions (reused launch after launch: all users
    are ordered on one stream)."""
    buf = _PART_WS.get(device)
    if buf is None or buf.numel() < elems:
        buf = torch.empty(max(elems, 1 << 22), device=device, dtype=torch.float32)
        _PART_WS[device] = buf
    return buf


class PartialItem(C.Structure):
    _fields_ = [("part", vp), ("out0", vp), ("out1", vp), ("nblk", i32), ("n", i32)]


PARTIAL_GROUP_MAX = 32


class PartialQueue:
    """Deferred second stage of the GroupNorm / LayerNorm parameter-gradient reductions (pdmk.h): each backward call leaves
    its per-block partials in a slab of its own; flush() sums up to 32 slabs per launch into dgamma / dbeta."""

    def __init__(self):
        self.items = []          # (slab tensor, dgamma, dbeta, nblk, n): the slab stays referenced until the flush

    def slab(self, device, nblk, n, dgamma, dbeta):
        if len(self.items) >= PARTIAL_GROUP_MAX:       # before the new slab joins: its kernel has not been launched yet
            self.flush()
        t = torch.empty(nblk * 2 * n, device=device, dtype=torch.float32)
        self.items.append((t, dgamma, dbeta, nblk, n))
        return t

    def flush(self):
        while self.items:
            chunk, self.items = self.items[:PARTIAL_GROUP_MAX], self.items[PARTIAL_GROUP_MAX:]
            arr = (PartialItem * len(chunk))()
            for a, (t, g0, g1, nblk, n) in zip(arr, chunk):
                a.part, a.out0, a.out1, a.nblk, a.n = _p(t), _p(g0), _p(g1), nblk, n
            _chk(_lib.pdmk_reduce_partials_group(C.cast(arr, vp), len(chunk), _st()), "pdmk_reduce_partials_group")


def _dims(fn, *args):
    a, b = i32(0), i32(0)
    _chk(fn(*args, C.byref(a), C.byref(b)), fn.__name__)
    return a.value, b.value


def groupnorm_bwd(x, dy, dx, gamma, beta, stats, dgamma, dbeta, ws, B, HW, Cc, ldx, lddy, lddx, G, gs, silu, acc, add=None,
                  queue=None):
    """queue: a PartialQueue - the dgamma / dbeta reduction is deferred to its next flush (one launch for many layers)."""
    if queue is not None:
        nblk, n = _dims(_lib.pdmk_groupnorm_bwd_partial_dims, B, HW, Cc, G, gs, dt(x))
        pw = queue.slab(x.device, nblk, n, dgamma, dbeta)
        dgamma = dbeta = None
    else:
        pw = part_ws(x.device, _ws_bytes(_lib.pdmk_groupnorm_bwd_part_workspace_bytes(G, gs)) // 4)
    _chk(_lib.pdmk_groupnorm_bwd(_p(x), _p(dy), _p(dx), _p(gamma), _p(beta), _p(stats), _p(dgamma), _p(dbeta), _p(ws),
                                 _p(pw), pw.numel(), B, HW, Cc, ldx, lddy, lddx, G, gs, int(silu), int(acc), _p(add),
                                 0 if add is None else add.stride(0), dt(x), _st()), "pdmk_groupnorm_bwd")


@_recordable("layernorm_fwd")
def layernorm_fwd(x, y, gamma, beta, stats, M, Cc, ldx, ldy, eps):
    _chk(_lib.pdmk_layernorm_fwd(_p(x), _p(y), _p(gamma), _p(beta), _p(stats), M, Cc, ldx, ldy, eps, dt(x), _st()),
         "pdmk_layernorm_fwd")


def layernorm_bwd(x, dy, dx, gamma, stats, dgamma, dbeta, M, Cc, ldx, lddy, lddx, acc, queue=None, add=None):
    """add: a second finished gradient of x ([M, Cc], any row stride) folded into the store of dx."""
    if queue is not None:
        nblk, n = _dims(_lib.pdmk_layernorm_bwd_partial_dims, M, Cc)
        pw = queue.slab(x.device, nblk, n, dgamma, dbeta)
        dgamma = dbeta = None
    else:
        pw = part_ws(x.device, _ws_bytes(_lib.pdmk_layernorm_bwd_part_workspace_bytes(M, Cc)) // 4)
    _chk(_lib.pdmk_layernorm_bwd(_p(x), _p(dy), _p(dx), _p(gamma), _p(stats), _p(dgamma), _p(dbeta), _p(pw),
                                 pw.numel(), M, Cc, ldx, lddy, lddx, int(acc), _p(add), 0 if add is None else add.stride(0),
                                 dt(x), _st()), "pdmk_layernorm_bwd")


@_recordable("attn_fwd")
def attn_fwd(q, k, v, o, lse, B, H, Nq, Nk, qs, ks, vs, os_, scale):
    """qs/ks/vs/os_ = (batch_stride, row_stride) in elements."""
    _chk(_lib.pdmk_attn_fwd(_p(q), _p(k), _p(v), _p(o), _p(lse), B, H, Nq, Nk, qs[0], qs[1], ks[0], ks[1], vs[0],
                            vs[1], os_[0], os_[1], scale, dt(q), _st()), "pdmk_attn_fwd")


def attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, B, H, Nq, Nk, qs, ks, vs, os_, dqs, dks, dvs, scale):
    nws = _ws_bytes(_lib.pdmk_attn_bwd_workspace_bytes(B, H, Nq, Nk)) // 4   # > 0: few keys, the dK/dV pass also splits the queries
    ws = torch.empty(nws, device=q.device, dtype=torch.float32) if nws else None
    _chk(_lib.pdmk_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(do), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), B, H, Nq,
                            Nk, qs[0], qs[1], ks[0], ks[1], vs[0], vs[1], os_[0], os_[1], dqs[0], dqs[1], dks[0],
                            dks[1], dvs[0], dvs[1], scale, _p(ws), 0 if ws is None else ws.numel(), dt(q), _st()),
         "pdmk_attn_bwd")


@_recordable("geglu_fwd")
def geglu_fwd(x, y, M, Fd, ldx, ldy, layout=0):
    """layout 0: x = [h | g] halves; 1: (h, g) interleaved in blocks of 8 columns (what EPI_GEGLU consumes)."""
    _chk(_lib.pdmk_geglu_fwd(_p(x), _p(y), M, Fd, ldx, ldy, int(layout), dt(x), _st()), "pdmk_geglu_fwd")


def geglu_bwd(x, dy, dx, M, Fd, ldx, lddy, lddx, layout=0):
    _chk(_lib.pdmk_geglu_bwd(_p(x), _p(dy), _p(dx), M, Fd, ldx, lddy, lddx, int(layout), dt(x), _st()), "pdmk_geglu_bwd")


_GEGLU_REFUSED = set()     # (M, N, K) the library has no fused GEGLU kernel for (learnt from eager calls: status -2)


def _launch_gemm_geglu(g, macs):
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = _lib.pdmk_gemm(C.byref(g), _st())
    if rc == -2:
        if not g.ln_gamma:
            _GEGLU_REFUSED.add((g.M, g.N, g.K))
        return False
    _chk(rc, "pdmk_gemm[geglu]")
    if PROFILE is not None:
        e1.record()
        PROFILE.append((("bf16", A_ROWK, B_ROWK, _lib.pdmk_gemm_last_candidate()), 2.0 * (macs if macs is not None else g.M * g.N * g.K),
                        e0, e1, (g.M, g.N, g.K, 1)))
    return True


def gemm_geglu(A, B, gl, f, M, N, K, lda, ldb, *, bias=None, macs=None, ln=None):
    """gl[M, N/2] = GEGLU(A @ B^T + bias) in the GEMM's epilogue, (hidden, gate) columns interleaved in blocks of 8; f (or
    None) receives the [M, N] pre-activation for the backward.  Returns False when the library has no fused kernel for
    the shape (status -2) - the caller then runs the projection and pdmk_geglu_fwd(layout=1) as two passes.
    While recording (lockstep forward) the answer must be known before the launch: shapes a previous eager call was refused
    for answer False at once, every other shape is recorded as fused (and raises at launch time if the library refuses)."""
    if (M, N, K) in _GEGLU_REFUSED:
        return False
    g = GemmArgs()
    g.A, g.B, g.C, g.bias = _p(A), _p(B), _p(gl), _p(bias)
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = lda, ldb, gl.stride(0)
    g.a_mode, g.b_mode, g.dtype = A_ROWK, B_ROWK, dt(A)
    g.splitk, g.alpha = 1, 1.0
    g.epilogue, g.C2, g.ldc2 = EPI_GEGLU, _p(f), 0 if f is None else f.stride(0)
    _set_ln(g, ln)
    if RECORD is not None:
        def run():
            if not _launch_gemm_geglu(g, macs):
                raise PdmkError(f"fused GEGLU refused for {(M, N, K)} during a recorded (lockstep) forward; run one eager "
                                f"forward first so that the shape is known")
        RECORD.append(Rec("gemm", run, g=g, macs=macs, keep=(A, B, gl, f, bias)))
        return True
    return _launch_gemm_geglu(g, macs)


def gemm_geglu_bwd(dy, wt, pre, dpre, M, N, K, lddy, ldwt, *, macs=None):
    """dpre[M, 2N] = gradient of the GEGLU pre-activation `pre` [M, 2N] (interleaved layout) given dy [M, K], the gradient of the
    Linear that consumed hidden * gelu(gate): (dy @ wt^T) pushed through GEGLU's backward in the GEMM's epilogue
    (PDMK_EPI_GEGLU_BWD).  Returns False when the library has no fused kernel for the shape (the caller then runs the plain
    input-gradient GEMM and pdmk_geglu_bwd)."""
    g = GemmArgs()
    g.A, g.B, g.C, g.C2 = _p(dy), _p(wt), _p(dpre), _p(pre)
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc, g.ldc2 = lddy, ldwt, dpre.stride(0), pre.stride(0)
    g.a_mode, g.b_mode, g.dtype = A_ROWK, B_ROWK, dt(dy)
    g.splitk, g.alpha = 1, 1.0
    g.epilogue = EPI_GEGLU_BWD
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = _lib.pdmk_gemm(C.byref(g), _st())
    if rc == -2:
        return False
    _chk(rc, "pdmk_gemm[geglu_bwd]")
    if PROFILE is not None:
        e1.record()
        PROFILE.append((("bf16", A_ROWK, B_ROWK, _lib.pdmk_gemm_last_candidate()), 2.0 * (macs if macs is not None else M * N * K),
                        e0, e1, (M, N, K, 1)))
    return True


@_recordable("quantize_e4m3")
def quantize_e4m3_(x):
    """In place: every element of the contiguous tensor x rounded to the nearest e4m3fn value (pdmk_quantize_e4m3)."""
    assert x.is_contiguous()
    _chk(_lib.pdmk_quantize_e4m3(_p(x), _p(x), x.numel(), dt(x), _st()), "pdmk_quantize_e4m3")


@_recordable("silu_fwd")
def silu_fwd(x, y):
    _chk(_lib.pdmk_silu_fwd(_p(x), _p(y), x.numel(), dt(x), _st()), "pdmk_silu_fwd")


def silu_bwd(x, dy, dx):
    _chk(_lib.pdmk_silu_bwd(_p(x), _p(dy), _p(dx), x.numel(), dt(x), _st()), "pdmk_silu_bwd")


@_recordable("copy2d")
def copy2d(src, dst, rows, cols, lds, ldd, accumulate=False):
    _chk(_lib.pdmk_copy2d(_p(src), _p(dst), rows, cols, lds, ldd, int(accumulate), dt(src), _st()), "pdmk_copy2d")


@_recordable("cast_permute")
def cast_permute(src, dst, n0, n1, n2, mode):
    _chk(_lib.pdmk_cast_permute(_p(src), _p(dst), n0, n1, n2, mode, dt(dst), _st()), "pdmk_cast_permute")


def colsum(x, out, rows, N, ld, accumulate=False, nbatch=1, ldo=0):
    _chk(_lib.pdmk_colsum(_p(x), _p(out), rows, N, ld, int(accumulate), nbatch, ldo, dt(x), _st()), "pdmk_colsum")


@_recordable("skinny_gemm")
def skinny_gemm(x, w, y, M, N, K, ldx, ldw, ldy, bias=None, accumulate=False):
    """y[M<=16, N] (+)= x @ w[N, K]^T + bias  (w in the compute dtype; x bf16/fp32; y fp32 or the compute dtype)."""
    _chk(_lib.pdmk_skinny_gemm(_p(x), dt(x), _p(w), _p(y), _p(bias), M, N, K, ldx, ldw, ldy, dt(w),
                               int(y.dtype == torch.float32), int(accumulate), _st()), "pdmk_skinny_gemm")


def skinny_wgrad(dy, x, dw, dbias, M, N, K, lddy, ldx, lddw):
    _chk(_lib.pdmk_skinny_wgrad(_p(dy), dt(dy), _p(x), _p(dw), _p(dbias), M, N, K, lddy, ldx, lddw, dt(x), _st()),
         "pdmk_skinny_wgrad")


def pool2x2_sum(src, dst, B, H, W, Cc):
    _chk(_lib.pdmk_pool2x2_sum(_p(src), _p(dst), B, H, W, Cc, dt(src), _st()), "pdmk_pool2x2_sum")


@_recordable("timestep_embed")
def timestep_embed(t, freqs, out, B, dim):
    _chk(_lib.pdmk_timestep_embed(_p(t), _p(freqs), _p(out), B, dim, dt(out), _st()), "pdmk_timestep_embed")


@_recordable("add_noise_velocity")
def add_noise_velocity(x0, noise, t, sa, sb, noisy, target, B, Cc, HW, cpad):
    _chk(_lib.pdmk_add_noise_velocity(_p(x0), _p(noise), _p(t), _p(sa), _p(sb), _p(noisy), _p(target), B, Cc, HW, cpad,
                                      dt(noisy), _st()), "pdmk_add_noise_velocity")


@_recordable("nchw_to_nhwc")
def nchw_to_nhwc(src, dst, B, Cc, HW, cpad):
    _chk(_lib.pdmk_nchw_to_nhwc(_p(src), _p(dst), B, Cc, HW, cpad, dt(dst), _st()), "pdmk_nchw_to_nhwc")


def embed_tokens(ids, tok, pos, out, ntok, T, D, vocab, ldt, ldp, ldo):
    _chk(_lib.pdmk_embed_tokens(_p(ids), _p(tok), _p(pos), _p(out), ntok, T, D, vocab, ldt, ldp, ldo, dt(out), _st()),
         "pdmk_embed_tokens")


def attn_fwd_causal(q, kk, v, o, lse, B, H, N, qs, ks, vs, os_, scale):
    """qs / ks / vs / os_ = (batch stride, row stride) in elements, as attn_fwd."""
    _chk(_lib.pdmk_attn_fwd_causal(_p(q), _p(kk), _p(v), _p(o), _p(lse), B, H, N, qs[0], qs[1], ks[0], ks[1], vs[0], vs[1],
                                   os_[0], os_[1], float(scale), dt(q), _st()), "pdmk_attn_fwd_causal")


def gelu_fwd(x, y):
    _chk(_lib.pdmk_gelu_fwd(_p(x), _p(y), x.numel(), dt(x), _st()), "pdmk_gelu_fwd")


def softmax_rows(s, p, rows, cols, lds, ldp):
    """p[r, :cols] = softmax(s[r, :cols]) for fp32 scores s; p in the compute dtype."""
    _chk(_lib.pdmk_softmax_rows(_p(s), _p(p), rows, cols, lds, ldp, dt(p), _st()), "pdmk_softmax_rows")


def latent_sample(moments, eps, latents, B, Cc, HW, ld, scale):
    _chk(_lib.pdmk_latent_sample(_p(moments), ld, _p(eps), _p(latents), B, Cc, HW, float(scale), dt(moments), _st()),
         "pdmk_latent_sample")


def nhwc_to_nchw(src, dst, B, Cc, HW, ld):
    _chk(_lib.pdmk_nhwc_to_nchw(_p(src), _p(dst), B, Cc, HW, ld, dt(src), _st()), "pdmk_nhwc_to_nchw")


def mse_fwd(a, b, w, out, slot, B, rows_per_b, cols, lda, ldb, scale):
    _chk(_lib.pdmk_mse_fwd(_p(a), dt(a), _p(b), dt(b), _p(w), _p(out), slot, B, rows_per_b, cols, lda, ldb,
                           float(scale), _st()), "pdmk_mse_fwd")


def mse_bwd(a, b, w, da, B, rows_per_b, cols, lda, ldb, ldda, gscale, accumulate):
    _chk(_lib.pdmk_mse_bwd(_p(a), dt(a), _p(b), dt(b), _p(w), _p(da), B, rows_per_b, cols, lda, ldb, ldda,
                           float(gscale), int(accumulate), _st()), "pdmk_mse_bwd")


def mse_fwd_bwd(a, b, w, out, slot, da, B, rows_per_b, cols, lda, ldb, ldda, scale, gscale, accumulate):
    """Loss value (out[slot] +=, skipped when out is None) and gradient seed (da, skipped when None) in one vectorised pass;
    falls back to the two scalar kernels when the shape is not 16-byte friendly."""
    ok = cols % 8 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 and (da is None or da.data_ptr() % 16 == 0)
    va, vb = (8 if a.dtype == torch.bfloat16 else 4), (8 if b.dtype == torch.bfloat16 else 4)
    ok = ok and lda % va == 0 and ldb % vb == 0 and (da is None or ldda % va == 0)
    if ok:
        _chk(_lib.pdmk_mse_fwd_bwd(_p(a), dt(a), _p(b), dt(b), _p(w), _p(out), slot, _p(da), B, rows_per_b, cols, lda, ldb,
                                   ldda, float(scale), float(gscale), int(accumulate), _st()), "pdmk_mse_fwd_bwd")
        return
    if out is not None:
        mse_fwd(a, b, w, out, slot, B, rows_per_b, cols, lda, ldb, scale)
    if da is not None:
        mse_bwd(a, b, w, da, B, rows_per_b, cols, lda, ldb, ldda, gscale, accumulate)


def axpby(x, y, alpha, beta):
    _chk(_lib.pdmk_axpby(_p(x), _p(y), float(alpha), float(beta), x.numel(), dt(x), _st()), "pdmk_axpby")


def adamw(p, g, m, v, n, lr, b1, b2, eps, wd, bias_corr, grad_scale, zero_grad, w_bf16=None):
    _chk(_lib.pdmk_adamw(_p(p), _p(g), _p(m), _p(v), n, _p(lr), b1, b2, eps, wd, _p(bias_corr), grad_scale,
                         int(zero_grad), _p(w_bf16), _st()), "pdmk_adamw")


def transpose_tiles(src, dst, table, ntiles):
    _chk(_lib.pdmk_transpose_tiles(_p(src), _p(dst), _p(table), ntiles, dt(src), _st()), "pdmk_transpose_tiles")


def conv_up2_supported(B, H, W, Ci, Co, dtype):
    """True when the library serves nearest-x2 upsample + 3x3 conv of a [B, H, W, Ci] image as four 2x2 phase convs."""
    return dtype == torch.bfloat16 and bool(_lib.pdmk_conv_up2_supported(B, H, W, Ci, Co, BF16))


def up2_pack_weights(w3, wp, wpt, Co, Ci):
    _chk(_lib.pdmk_up2_pack_weights(_p(w3), _p(wp), _p(wpt), Co, Ci, dt(wp), _st()), "pdmk_up2_pack_weights")


def up2_combine_wgrad(dwp, dw3, Co, Ci):
    _chk(_lib.pdmk_up2_combine_wgrad(_p(dwp), _p(dw3), Co, Ci, _st()), "pdmk_up2_combine_wgrad")


def sumsq(x, n, out, slot):
    _chk(_lib.pdmk_sumsq(_p(x), n, _p(out), slot, _st()), "pdmk_sumsq")
