import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def locked_make(directory, target):
    """`make -s -C directory target` under a host-wide lock: the suite runs on several xdist workers (pytest.ini) and two of them must not rebuild the same library at once"""
    import fcntl
    import subprocess
    with open("/tmp/vvcx_tests_make.lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            subprocess.check_call(["make", "-s", "-C", directory, target])
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # libvvcx.so links the system HIP runtime, the PyTorch wheel carries its own: PyTorch has to initialise its device context first
    # (INTEGRATION.md "Using the library next to PyTorch in one process"), whatever the order in which test modules load the library
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            # also in the xdist controller process, which runs no test itself: the product library is what this suite exercises (it raises if the HIP build is missing)
            import importlib
            importlib.import_module("reduce-complexity-for-intra-coding-of-vvc_amd").load_library(None)
    except Exception:
        pass
