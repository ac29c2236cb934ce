import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# Load torch's HIP runtime BEFORE the C-ABI library, exactly like bench.py does: both ship a libamdhip64.so.7 / librccl.so.1
# and the dynamic loader keeps whichever comes first, so fixing the order keeps test and bench processes identical.
try:
    import torch  # noqa: F401
except Exception:      # pragma: no cover
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present() -> bool:
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; GPU tests fail loudly (no skip, no fallback) when it is missing."""
    import velo_amd  # noqa: F401
    from velo_amd import api
    return api.load_library()


@pytest.fixture(scope="session")
def diag_lib(hip_lib):
    """The -DVELO_DIAGNOSTICS build of the same source: the only build that honours the A/B environment switches (kernel variants,
    grid shapes, two-launch LM iterations ...).  The product library ignores them, so the tests that sweep variants create their
    contexts on this one: api.Context(0, lib=diag_lib).  Loaded AFTER the product library (both are built with -Bsymbolic)."""
    import velo_amd  # noqa: F401
    from velo_amd import api
    return api.load_diagnostics_library()


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU is a usage error we want to see, so nothing is skipped silently here.
    return
