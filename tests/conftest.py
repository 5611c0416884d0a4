import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def port():
    """Plain-C restatement of the reference (oracle/lte_oracle.c)."""
    from oracle import pyoracle
    return pyoracle.port()


@pytest.fixture(scope="session")
def ref():
    """The reference itself, compiled in place; skipped where oracle/_ref is not available."""
    from oracle import pyoracle
    L = pyoracle.ref()
    if L is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return L


@pytest.fixture(scope="session")
def ref_phy(ref):
    phy = ref.ref_phy_new(4, 17, 1, 100)  # 30.72 MHz, cell 17, 1 port, 100 RB
    yield phy
    ref.ref_phy_free(phy)


@pytest.fixture(scope="session")
def ctx():
    """GPU context.  No skip-on-missing-library: on the GPU box a missing HIP extension must fail loudly."""
    import openlte_amd
    c = openlte_amd.Context(0)
    yield c
    c.close()
