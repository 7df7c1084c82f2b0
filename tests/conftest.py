import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def helm_lib():
    """Build (if stale) and load libhelm.so; GPU tests call through this C ABI."""
    import __graft_entry__ as g
    g.build()
    from zephyr_amd import _lib
    return _lib.load()
