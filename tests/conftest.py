import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLD, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def gold_bn():
    return load_golden("fn_bn.npz")


@pytest.fixture(scope="session")
def gold_sam():
    return load_golden("fn_sam.npz")


@pytest.fixture(scope="session")
def gold_tail():
    return load_golden("fn_tail.npz")


@pytest.fixture(scope="session")
def gold_small():
    return load_golden("net_small.npz")


@pytest.fixture(scope="session")
def gold_full():
    return load_golden("net_full.npz")


@pytest.fixture(scope="session")
def gold_input():
    return load_golden("fn_input.npz")


@pytest.fixture(scope="session")
def gold_ccl():
    return load_golden("fn_ccl.npz")


@pytest.fixture(scope="session")
def synth_params_np():
    from diga_amd import synth
    return synth.synth_params(0)
