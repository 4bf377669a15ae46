import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    import oracle_binding
    return oracle_binding.rt


@pytest.fixture(scope="session")
def ob():
    import oracle_binding
    oracle_binding.lib()
    return oracle_binding


_SCENES = {}


@pytest.fixture(scope="session")
def scenes(rt):
    """Lazy cache of host scenes by file name."""
    def get(name, **kw):
        key = (name, tuple(sorted(kw.items())))
        if key not in _SCENES:
            _SCENES[key] = rt.Scene.load(name, **kw)
        return _SCENES[key]
    return get


@pytest.fixture(scope="session")
def gpu(rt):
    """One rt_context on device 0 for the whole GPU session (fails loudly without the HIP library)."""
    r = rt.Renderer(0)
    yield r
    r.close()
