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


@pytest.fixture(autouse=True)
def _poisoned_allocator(request):
    """GPU tests: whatever torch's caching allocator hands out next is full of NaN, not of a previous test's results --
    an output buffer that a kernel or the one-call driver forgets to write shows up as NaN instead of passing by luck
    (it did once: the dictionary's dry mix of a group without reverb)."""
    if request.node.get_closest_marker('gpu') is not None:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            junk = [torch.full((n,), float('nan'), device='cuda') for n in (1 << 24, 1 << 22, 1 << 22, 1 << 20, 1 << 20,
                                                                           1 << 18, 1 << 18, 1 << 16, 1 << 16, 1 << 14)]
            torch.cuda.synchronize()
            del junk
    yield
