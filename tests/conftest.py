import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cpu_workers(config):
    """Worker processes for a CPU-only run (pytest-xdist, no -n given): the emulator parity tests are ~10^3 x slower than the GPU's, and a plain
    `pytest tests -m "not gpu"` on the 8-core build container took 27 minutes serially.  Never on a box with a GPU (one device: the -m gpu tests run in
    one process), never inside a worker, FS_TEST_WORKERS=0 turns it off / =N pins it."""
    if hasattr(config, "workerinput") or not config.pluginmanager.hasplugin("xdist"):
        return 0
    if getattr(config.option, "numprocesses", None) is not None or config.getoption("collectonly", False) or config.getoption("usepdb", False):
        return 0
    env = os.environ.get("FS_TEST_WORKERS")
    if env is not None:
        return max(0, int(env))
    try:
        import torch
        if torch.cuda.is_available():
            return 0
    except Exception:
        pass
    return max(0, min(4, (os.cpu_count() or 1) // 2))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    n = _cpu_workers(config)
    if n > 1:
        # (xdist's own pytest_configure is trylast: it sees these values and starts the workers; the emulator library is built ONCE, here, before they exist)
        from tests import emu_lib
        emu_lib.build_emu()
        config.option.numprocesses = n
        config.option.dist = "load"
        config.option.tx = ["popen"] * n


@pytest.fixture(scope="session")
def repo_root():
    return ROOT
