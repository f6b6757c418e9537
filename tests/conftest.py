import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: opt-in long runs (skipped unless their environment variable is set)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        # a plain dict: an NpzFile inflates an array again on EVERY g["..."] access (the glue traces are indexed once per step)
        with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
            return {k: z[k] for k in z.files}

    return load
