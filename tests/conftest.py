import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "atc-reinforcement-learning_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


# the sector compiler's on-disk grid cache (atc_hip/scenario.py:build_grid_cached) stays inside a per-session temporary
# directory: tests never write to the user's ~/.cache
_OWN_CACHE = None
if "ATC_HIP_CACHE" not in os.environ:
    import tempfile
    _OWN_CACHE = os.environ["ATC_HIP_CACHE"] = tempfile.mkdtemp(prefix="atc_hip_cache_")


def pytest_sessionfinish(session, exitstatus):
    if _OWN_CACHE:
        import shutil
        shutil.rmtree(_OWN_CACHE, ignore_errors=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """No GPU test may hang the box (a resident step server, a launch that never returns): 15 minutes each unless it names its own."""
    import pytest
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900))


def pytest_sessionstart(session):
    """Compile the native pieces when a fresh checkout has none (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as entry
    entry.ensure_built()
