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


def _hip_device_present():
    if not os.path.exists("/dev/kfd"):
        return False
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a HIP device skips the gpu-marked tests instead of erroring in Engine creation."""
    if _hip_device_present():
        return
    skip = pytest.mark.skip(reason="needs an MI355X (no HIP device on this box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


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


_SESSION_T0 = __import__("time").time()


def parity_log(**record):
    """GPU parity tests append what they MEASURED (not only that it passed) to gpurun_out/parity/<session start>.jsonl -- one file per
    pytest session, so that partial re-runs on another box add to the earlier records instead of replacing them when gpurun merges the
    directory back; tools/make_parity_report.py turns the directory into profiles/parity_rNN.md (latest record per case)."""
    import json
    import time
    out = os.path.join(ROOT, "gpurun_out", "parity")
    os.makedirs(out, exist_ok=True)
    record["_t"] = time.time()
    with open(os.path.join(out, "%d_%d.jsonl" % (int(_SESSION_T0), os.getpid())), "a") as f:
        f.write(json.dumps({k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in record.items()}) + "\n")


@pytest.fixture(scope="session")
def chairs32_decoder():
    """The second fixture decoder: 32-D codes, fitted to a different (taller) shape family (tools/fit_decoder_gpu.py
    --name chairs32 --code-len 32 --half 0.36 0.55 0.36); oracle form."""
    from dsp_slam_amd import fixtures
    from oracle import dsp_oracle
    return dsp_oracle.fold_decoder(fixtures.load_decoder_npz(fixtures.fixture_path("chairs32")), fixtures.fixture_specs("chairs32"))


def have_complex_fixture():
    from dsp_slam_amd import fixtures
    return os.path.exists(fixtures.fixture_path("complex")) and os.path.exists(os.path.join(GOLDEN, "golden_recon_complex.npz"))


@pytest.fixture(scope="session")
def complex_decoder():
    """The third fixture decoder (round 5): fitted to a non-convex multi-part shape family whose parameters depend on all 64 code dimensions
    (synth.complex_car_sdf; tools/fit_decoder_gpu.py --shape complex); oracle form."""
    from dsp_slam_amd import fixtures
    from oracle import dsp_oracle
    if not have_complex_fixture():
        pytest.skip("tests/golden/decoder_complex.npz not generated")
    return dsp_oracle.fold_decoder(fixtures.load_decoder_npz(fixtures.fixture_path("complex")), fixtures.fixture_specs("complex"))
