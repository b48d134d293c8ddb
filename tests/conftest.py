import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:      # helper modules beside the tests (rot_cond, precision_model)
    sys.path.insert(1, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


_CONFIG = None


def pytest_configure(config):
    global _CONFIG
    _CONFIG = config
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Multi-rank GPU tests need FRESH rank processes.  Start the multiprocessing fork server now -- before anything in this
    # process initialises the GPU (torch.cuda.is_available() below already does) -- so that ranks are forked from a
    # process that never touched it, instead of fork+exec'ing out of a GPU-initialised pytest process.
    import multiprocessing.forkserver as fs
    fs.ensure_running()


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


def pytest_runtest_logfinish(nodeid, location):
    """Flush the progress output after every test: with stdout redirected to a file or a pipe Python block-buffers it, and a GPU run
    that writes nothing for minutes is taken to be hung by the pool's watchdog (the 5-minute GPU suite was killed twice that way)."""
    import sys
    try:
        if _CONFIG is not None:
            _CONFIG.get_terminal_writer().flush()
        sys.stdout.flush()
        sys.stderr.flush()
    except Exception:
        pass
