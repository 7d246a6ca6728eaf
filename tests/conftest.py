import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def klib():
    """The ctypes binding of libktlvo_hip.so (import fails loudly if the library is missing)."""
    import slam_amd.lib as L
    return L


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    return O


@pytest.fixture()
def make_ctx(klib):
    made = []

    def _make(**kw):
        # The parity tests compare with the oracle bit for bit, which is the order-independent reading of LK's window sums
        # (an option of the library; its default is the float tree, which the tests that mean it ask for by name).
        kw.setdefault("lk_sums", klib.LK_SUMS_EXACT)
        ctx = klib.Context(klib.default_config(**kw))
        made.append(ctx)
        return ctx

    yield _make
    for c in made:
        c.close()
