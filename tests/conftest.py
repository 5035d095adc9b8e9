import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "atc-reinforcement-learning_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Compile the native pieces when a fresh checkout has none (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as entry
    entry.ensure_built()
