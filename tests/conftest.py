import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def _cpu_threads(workers):
    """intra-op threads per test process.  The oracle passes of the parity tests are 64 x 64-tile networks: on the GPU
    box's 256 host threads torch's default (one OpenMP thread per core) spends its time in barriers - the round-2 suite
    took 1001 s that way, 103 s with 6 workers x 20 threads (gpurun_out/r03_t0.log)."""
    n = os.cpu_count() or 8
    return max(4, min(20, n // max(workers, 1)))


def _workers(config):
    env = os.environ.get("XV2_TEST_WORKERS")
    if env is not None:
        return max(0, int(env))
    if getattr(config.option, "numprocesses", None) is not None:      # -n given on the command line
        return 0
    expr = getattr(config.option, "markexpr", "") or ""
    if "gpu" not in expr or "not gpu" in expr:
        return 0
    if (os.cpu_count() or 1) < 32 or not _gpu_present():
        return 0
    return 6


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """`pytest -m gpu` on the GPU box: spread the parity tests over 6 xdist workers (each with its own HIP context on
    the one GPU and a bounded OpenMP team for its oracle passes).  xdist's own pytest_cmdline_main runs last and turns
    numprocesses into worker specs.  XV2_TEST_WORKERS=0 keeps everything in one process."""
    if not config.pluginmanager.hasplugin("xdist") or os.environ.get("PYTEST_XDIST_WORKER"):
        return None
    n = _workers(config)
    if n > 0:
        config.option.numprocesses = n
        os.environ.setdefault("OMP_NUM_THREADS", str(_cpu_threads(n)))
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("OMP_NUM_THREADS") is None and (os.cpu_count() or 1) >= 32:
        # single-process runs on a many-core host: same cap (set before torch spins up its thread pool)
        os.environ["OMP_NUM_THREADS"] = str(_cpu_threads(1))
    try:
        import torch
        want = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
        if want > 0 and torch.get_num_threads() > want:
            torch.set_num_threads(want)
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
