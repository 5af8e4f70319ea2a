import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # libvvcx.so links the system HIP runtime, the PyTorch wheel carries its own: PyTorch has to initialise its device context first
    # (INTEGRATION.md "Using the library next to PyTorch in one process"), whatever the order in which test modules load the library
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
