"""pytest configuration: markers, paths, shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    """Load one committed fixture (data only; numpy.load never unpickles here)."""
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def synthetic_case(name):
    """Golden outputs + the seeded inputs they were generated from."""
    from oracle import gp_oracle
    g = load_golden(name)
    inputs, testing, theta, invQ, invQt = gp_oracle.benchmark_inputs(
        int(g["seed"]), int(g["N"]), int(g["D"]), int(g["M"]))
    g.update(inputs=inputs, testing=testing, theta=theta, invQ=invQ, invQt=invQt)
    return g


SYNTHETIC_CASES = ["c1_n100_d5", "bench_n250_d10", "c2_n250_d11", "c4_n300_d11",
                   "c5_n300_d16", "odd_n37_d3", "one_n1_d1"]


@pytest.fixture(scope="session")
def gpu_lib():
    """The C-ABI library, loaded; fails (never skips) if the GPU path is absent."""
    from gp_emulator_amd import _lib
    return _lib.load()
