"""pytest configuration: `gpu` marker, shared fixtures (oracle / product libraries)."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU comparison")


@pytest.fixture(scope="session")
def oracle():
    from criteria3d_amd import build
    from tests import checkers
    if not checkers.ORACLE_LIB.exists():
        build.build_oracle(with_reference=False)
    return checkers.load_oracle()


@pytest.fixture(scope="session")
def product():
    """The HIP library.  Fails (never skips, never falls back) when it is missing."""
    from criteria3d_amd import build, capi
    build.build_product()
    return capi.load_product()
