"""The LDL^T solves k_solve<0> (packed) and k_solve<2> (rows in lanes, default) of gn_kernels.hip, emulated on the CPU (tests/solve_emulator.py), against numpy -- on the reference's own
recorded normal equations (tests/golden/golden_recon_*.npz: it_H, it_b, it_dx) and on synthetic SPD systems, 71 x 71 and 6 x 6."""
import numpy as np
import pytest

import solve_emulator as S
from conftest import golden


def test_packed_index_mapping_is_a_bijection_on_the_lower_triangle():
    seen = set()
    for e in range(S.LDL_NP):
        i, j = S.packed_to_ij(e)
        assert 0 <= j <= i < S.NS1
        assert e == j * S.NS1 - j * (j - 1) // 2 + (i - j)
        seen.add((i, j))
    assert len(seen) == S.LDL_NP == 2628 and S.LDL_EPT == 6


@pytest.mark.parametrize("name", ["golden_recon_cfg2.npz", "golden_recon_cfg5.npz", "golden_recon_small.npz"])
def test_ldl_on_the_references_recorded_systems(name):
    g = golden(name)
    for e in range(g["it_H"].shape[0]):
        H, b = g["it_H"][e].astype(np.float64), g["it_b"][e].astype(np.float64)
        dx, sing = S.ldl_solve(H, b)
        assert not sing
        ref = np.linalg.solve(H, b)
        assert np.abs(dx - ref).max() <= 1e-9 * np.abs(ref).max()
        dxb, singb = S.solve_rows_in_lanes(H, b)         # k_solve<2>: same pivots, rows above them eliminated too
        assert not singb and np.abs(dxb - ref).max() <= 1e-9 * np.abs(ref).max() and np.abs(dxb - dx).max() <= 1e-10 * np.abs(ref).max()
        # and the reference's own float32 torch.inverse(H) @ b is within ITS round-off of both
        assert np.abs(dx - g["it_dx"][e]).max() <= 2e-3 * np.abs(ref).max()


def test_ldl_small_and_indefinite():
    rng = np.random.default_rng(0)
    J = rng.normal(size=(40, 6))
    H = J.T @ J / 40 + 1e-2 * np.eye(6)
    b = rng.normal(size=6)
    dx, sing = S.ldl_solve(H, b)
    assert not sing and np.allclose(dx, np.linalg.solve(H, b), rtol=1e-11, atol=1e-13)
    dxb, singb = S.solve_rows_in_lanes(H, b)         # 6 x 6 inside the 72 x 72 grid: everything beyond index 6 is NaN garbage that stays put
    assert not singb and np.allclose(dxb, dx, rtol=1e-11, atol=1e-13)
    H[2, 2] = -1.0                                   # not positive definite: reported, as the Gauss-Jordan kernel reports a zero pivot
    assert S.ldl_solve(H, b)[1] and S.solve_rows_in_lanes(H, b)[1]
    H = np.full((6, 6), np.nan)
    assert S.ldl_solve(H, b)[1]


def test_ldl_full_size_random_spd():
    rng = np.random.default_rng(1)
    J = rng.normal(size=(500, 71)) * rng.uniform(0.01, 30.0, size=71)
    H = J.T @ J / 500 + np.diag(np.r_[np.ones(7), 0.25 * np.ones(64)])
    b = rng.normal(size=71)
    dx, sing = S.ldl_solve(H, b)
    ref = np.linalg.solve(H, b)
    assert not sing and np.abs(dx - ref).max() <= 1e-10 * np.abs(ref).max()
    dxb, singb = S.solve_rows_in_lanes(H, b)
    assert not singb and np.abs(dxb - ref).max() <= 1e-10 * np.abs(ref).max()
