import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))

    return load


@pytest.fixture(autouse=True)
def _oracle_decoder_keeps_keys_and_values(request):
    """The -m gpu tests decode hundreds of prefixes with the CPU oracle (twice: on the fp32 path and on the HIP path's own
    prefixes).  They use the oracle's key/value-cached greedy loop -- the same sums per position, held to the reference's
    golden ids by tests/test_oracle_golden.py::test_decoder_ids_bit_exact[*-True] -- so the suite takes minutes, not a
    quarter of an hour.  The CPU tests keep the reference's literal full re-forward."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from oracle import patchioner_oracle as O
    old = O.DeCapOracle.fast
    O.DeCapOracle.fast = True
    yield
    O.DeCapOracle.fast = old
