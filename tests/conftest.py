import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def golden_cases(npz, kind):
    """Group 'kind_NN_field' entries of primitives.npz into dicts."""
    cases = {}
    pre = kind + "_"
    for k in npz.files:
        if not k.startswith(pre):
            continue
        rest = k[len(pre):]
        idx, field = rest.split("_", 1)
        if not idx.isdigit():
            continue
        cases.setdefault(int(idx), {})[field] = npz[k]
    return [cases[i] for i in sorted(cases)]


@pytest.fixture(scope="session")
def primitives():
    return load_golden("primitives")


FUNC_CASES = ["func_64_s1", "func_64_s2_ties", "func_96x80_s5", "func_96x80_s6",
              "func_128_s7_ct3", "func_256_s9", "func_256_s10_ties"]
