import os
import sys

# The CPU oracles are OpenMP code with thousands of short parallel regions.  libgomp's default lets an idle thread spin for ~300k
# iterations: on a host whose cores are shared (CI, a GPU box that grants 16 of 256 hardware threads) that turned a 1-second oracle test
# into minutes; sleeping at once (OMP_WAIT_POLICY=PASSIVE) costs a futex round trip per region instead (the suite: 25 s -> 144 s).  A short
# spin, set before anything loads libgomp, is fast when the cores are free and harmless when they are not.
os.environ.setdefault("GOMP_SPINCOUNT", "20000")

import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """The parity tests must exercise the in-tree HIP library: an environment that redirects the binding is refused."""
    if os.environ.get("TEEFLOW_LIB") and any(it.get_closest_marker("gpu") for it in items):
        raise pytest.UsageError("TEEFLOW_LIB is set: -m gpu tests run only against tee_optical_flow_amd/libteeflow_hip.so "
                                "(the override exists for A/B builds of the product in tools/, never for tests)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure). Built on demand with gcc."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def engine():
    """One DenseFlow handle on cuda:0 for the whole GPU session (fails loudly without the HIP library)."""
    import tee_optical_flow_amd as T
    eng = T.DenseFlow(device_id=0)
    yield eng
    eng.close()

