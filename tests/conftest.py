import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU (/dev/kfd absent)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="module", params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def api(request):
    """C-ABI binding under test: host-emulated kernels (CPU suite) or the real gfx950 library."""
    if request.param == "emu":
        from tests.emu import emu_api
        return emu_api()
    from searcharray_amd import _lib
    return _lib.api()


@pytest.fixture(scope="module")
def default_api(api):
    """Make `api` the package default for code that does not take an explicit binding
    (SearchArray); restored afterwards."""
    from searcharray_amd import _lib
    old = _lib._api
    _lib.use_api(api)
    yield api
    _lib.use_api(old)


@pytest.fixture
def on_emu(request):
    """True when the test runs on the host-emulated kernels (CPU suite): the heaviest parametrisations are thinned
    there -- the GPU run keeps all of them."""
    cs = getattr(request.node, "callspec", None)
    return bool(cs and cs.params.get("api") == "emu")


@pytest.fixture(autouse=True)
def _option_scope():
    """library options set by a test (tests.helpers.set_opt) end with it"""
    from searcharray_amd import options
    from tests import helpers
    helpers._scope = options.Scope()
    yield
    helpers._scope.close()
    helpers._scope = None
