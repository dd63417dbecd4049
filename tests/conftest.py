import json
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


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle_py import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def si128():
    d = np.load(os.path.join(GOLDEN, "data_si128.npz"))
    return np.ascontiguousarray(d["X"]), np.ascontiguousarray(d["y"])


@pytest.fixture(scope="session")
def sine():
    d = np.load(os.path.join(GOLDEN, "data_sine_4160.npz"))
    return np.ascontiguousarray(d["X"]), np.ascontiguousarray(d["y"])


@pytest.fixture(scope="session")
def golden_si128():
    return load_json("golden_si128.json")


@pytest.fixture(scope="session")
def golden_sine():
    return load_json("golden_sine.json")


@pytest.fixture(scope="session")
def ref_log():
    return load_json("ref_log_si128.json")


def synth(n, d=10, seed=15618, scale=10.0, noise=0.1):
    """Synthetic workload of the reference's shape: X ~ U(-scale,scale)^d, y = sin(x0)+noise."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(-scale, scale, (n, d))
    y = np.sin(X[:, 0]) + noise * rng.standard_normal(n)
    return np.ascontiguousarray(X), np.ascontiguousarray(y)


HP_DEFAULT = [0.5, 0.5, 0.5]
HP_BCM = [1.5, 1.5, 1.5]
HP_DENSE = [3.762111, -1.152105, -0.384461]
