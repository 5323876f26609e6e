import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def lib():
    from ddsp_piano_amd import _lib
    return _lib.load()


@pytest.fixture(autouse=True)
def _options_follow_the_environment():
    """Tests flip DDSPP_* switches with util.set_option (environment + reload).  monkeypatch restores the environment
    at teardown; this fixture (set up first, torn down last) then makes the package re-read it, so no test inherits
    another one's switches."""
    yield
    from ddsp_piano_amd import _lib
    _lib.options.reload()
