import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the dense-front kernel gets a launch of its own from this many qualifying workgroups on (8192 in production): low enough
    # for the test matrices to reach both the dedicated launch and the path inside the general launch
    os.environ.setdefault("PANGULU_HIP_FRONT_MIN_WGS", "64")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """The shared objects are build products (git-ignored); build them once if this checkout has none."""
    from pangulu_amd import _lib

    need = [_lib.library_path("r64"), os.path.join(ROOT, "oracle", "_build", "libpangulu_oracle_r64.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__

        __graft_entry__.build()
    yield
