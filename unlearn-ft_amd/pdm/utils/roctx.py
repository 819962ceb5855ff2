"""Phase markers of the training step: roctx ranges (rocprofv3 --marker-trace shows them next to the kernel timeline) and,
when a measurement asks for it, HIP-event timing of the same ranges.

The reference has no tracing at all; the north_star's evidence clause needs a per-phase breakdown that does
not depend on guessing phases from kernel symbol names.  Range names: fwd_teacher, fwd_student, loss, bwd, allreduce,
adamw (+ diffuse, wt_refresh).  The roctx library is loaded only with PDMK_ROCTX=1 (a training process does not pull a
profiler SDK library in by default; the HIP-event timing of the same ranges - bench.py's `extras.phases_ms_*` - works
without it).  Run `PDMK_ROCTX=1 rocprofv3 --kernel-trace --marker-trace -- python3 bench.py ...` to see the ranges on the
kernel timeline; keep it off for `--pmc` passes.
"""
import ctypes
import os

import torch

_lib = None
if os.environ.get("PDMK_ROCTX", "0") == "1":
    for _name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
        try:
            _lib = ctypes.CDLL(_name)
            _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            _lib.roctxRangePushA.restype = ctypes.c_int
            _lib.roctxRangePop.restype = ctypes.c_int
            break
        except (OSError, AttributeError):
            _lib = None

PHASE_LOG = None     # a list: every range is then also bracketed by HIP events on the current stream (eager runs only)


class phase:
    __slots__ = ("name", "e0")

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _lib is not None:
            _lib.roctxRangePushA(self.name.encode())
        self.e0 = None
        if PHASE_LOG is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PHASE_LOG.append((self.name, self.e0, e1))
        if _lib is not None:
            _lib.roctxRangePop()
        return False


def summarize(log):
    """{phase: milliseconds} from a PHASE_LOG list (call after a device synchronize)."""
    out = {}
    for name, e0, e1 in log:
        out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
    return {k_: round(v, 3) for k_, v in out.items()}
