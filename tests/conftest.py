import os
import sys

import pytest

os.environ.setdefault("PDMK_ENV_DYNAMIC", "1")     # lets tests force GEMM candidates through PDMK_RING_CFG / PDMK_WGRAD_CFG
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "unlearn-ft_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle is the slow half of the GPU suite: torch picks one thread per hardware thread of the HOST (128 on the
    # GPU boxes) while a one-GPU job owns 16 cores - the full-size oracle step takes 21.5 s at 128 threads and 3.9 s at 16
    # (profiles/r02_cpu_baseline_full.json, profiles/r02_bench.json)
    import torch
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _gpu_test_hygiene(request):
    """Between GPU tests: collect garbage and drain the device - housekeeping only.  Several tests build hipGraphs whose owners
    sit in reference cycles; collecting them here, with the device idle, keeps graph destruction out of the middle of a later
    test's replays and the suite's memory footprint flat.  (Round 2's host segfault inside hipGraphLaunch, which this fixture was
    first written against, is root-caused and removed by construction - an unbounded walk over `parallel_streams_` in
    hip::Graph::UpdateStreams that only multi-branch graphs enter; every captured graph is single-stream now)"""
    gpu = request.node.get_closest_marker("gpu") is not None
    if gpu:
        import gc
        import torch
        if torch.cuda.is_available():
            gc.collect()
            torch.cuda.synchronize()
    yield
    if gpu:
        import gc
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            gc.collect()
