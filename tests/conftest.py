import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np
    here = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(here, "cases.json")) as f:
        cases = json.load(f)
    arr = np.load(os.path.join(here, "cases.npz"))
    return cases, arr


@pytest.fixture(scope="session")
def golden_lowrank():
    """tests/golden/lowrank.{json,npz}: low-rank mode, the random objects of every case included (make_golden_lowrank.py)."""
    import json
    import types
    import numpy as np
    here = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(here, "lowrank.json")) as f:
        cases = json.load(f)
    arr = np.load(os.path.join(here, "lowrank.npz"))

    def sketches(name, num_levels):
        out = []
        for i in range(num_levels - 1):
            k1, k2, r = (int(v) for v in arr[f"{name}/sk{i}/shape"])
            out.append(types.SimpleNamespace(k1=k1, k2=k2, r=r, **{k: arr[f"{name}/sk{i}/{k}"] for k in ("colptr", "i1", "i2", "val")}))
        return out
    return cases, arr, sketches
