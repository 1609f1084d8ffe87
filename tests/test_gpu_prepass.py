"""GPU: the low-precision prepass kernel (mlp_lp_kernel.hip, through dsp_decode_sdf_prepass) against the direct statement
of its arithmetic and against the fp32 oracle.

The prepass is never a result -- it only classifies samples whose occupancy is exactly 0 or 1 -- so what matters is
(a) it computes what it says (same rounding points as lp_emulator.reference_forward: only accumulation order differs) and
(b) its distance to the fp32 decoder stays inside the margin the optimiser uses (DSP default delta: 4x the measured max)."""
import numpy as np
import pytest

from oracle import dsp_oracle as O
from dsp_slam_amd import engine as E, _lib as L
import lp_emulator as LE

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(oracle_decoder):
    e = E.Engine(oracle_decoder.layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    yield e
    e.close()


@pytest.mark.parametrize("dtype,tol_lp,tol_fp32", [(L.PREPASS_F16, 1.5e-4, 4e-4), (L.PREPASS_BF16, 1e-3, 4e-3)])
@pytest.mark.parametrize("n", [1, 31, 32, 33, 127, 128, 129, 1000, 40000])
def test_prepass_decode(eng, oracle_decoder, dtype, tol_lp, tol_fp32, n):
    rng = np.random.default_rng(7 * n + dtype)
    code = (rng.normal(size=64) * 0.2).astype(np.float32)
    pts = rng.uniform(-1.0, 1.0, size=(n, 3)).astype(np.float32)
    got = eng.decode_sdf_prepass(code, pts, dtype)
    ref32 = O.decode_sdf(oracle_decoder, code, pts)
    want = LE.reference_forward(oracle_decoder, code, pts, dtype)
    assert got.shape == (n,)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() < tol_lp, np.abs(got - want).max()
    assert np.abs(got - ref32).max() < tol_fp32, np.abs(got - ref32).max()


def test_prepass_matches_fp32_kernel_closely(eng):
    """device fp32 kernel vs device prepass on a large sample: the number the default margin is derived from."""
    rng = np.random.default_rng(3)
    worst = {}
    for trial in range(4):
        code = (rng.normal(size=64) * (0.1 + 0.1 * trial)).astype(np.float32)
        u = rng.normal(size=(60000, 3))
        pts = (u / np.linalg.norm(u, axis=1, keepdims=True) * rng.uniform(0, 1, size=(60000, 1)) ** (1 / 3)).astype(np.float32)
        ref = eng.decode_sdf(code, pts)
        for name, dt in (("f16", L.PREPASS_F16), ("bf16", L.PREPASS_BF16)):
            worst[name] = max(worst.get(name, 0.0), float(np.abs(eng.decode_sdf_prepass(code, pts, dt) - ref).max()))
    print("prepass max |sdf_lp - sdf_fp32| over 240k unit-ball points:", worst)
    assert worst["f16"] < 4e-4 and worst["bf16"] < 4e-3
