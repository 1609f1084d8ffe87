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


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def cars_state_dict():
    from dsp_slam_amd import fixtures
    return fixtures.load_decoder_npz(fixtures.fixture_path("cars"))


@pytest.fixture(scope="session")
def oracle_decoder(cars_state_dict):
    from dsp_slam_amd import fixtures
    from oracle import dsp_oracle
    return dsp_oracle.fold_decoder(cars_state_dict, fixtures.SPECS)
