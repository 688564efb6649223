import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # The GPU box shows every host core in the affinity mask but grants a 16-CPU share: let the CPU oracle use that share, no more
    # (PyTorch would otherwise start one thread per visible core and crawl).
    import torch
    from tests import helpers
    torch.set_num_threads(helpers.cpu_threads())
