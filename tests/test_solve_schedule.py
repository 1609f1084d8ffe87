"""k_solve's two schedules of the same elimination, emulated lane for lane in numpy (CPU).

dsp_slam_amd/csrc/gn_kernels.hip solves the 71 x 71 (pose-only: 6 x 6) normal equations pivot-free with rows in lanes: wave w < 9 holds
columns 8w .. 8w+7 of the augmented matrix for rows 0 .. 63 in eight registers (v0[jj], lane = row) and rows 64 .. 71 in one more
(vx: lane l = row 64 + (l & 7) of column 8w + (l >> 3)); rows above the pivot are eliminated too, so column n ends as d_i dx_i.
SOLVER 2 (round 4) publishes one column and passes one workgroup barrier per pivot; SOLVER 3 (round 5, default) lets wave kb run the eight
steps of its panel on its own registers, publish each column as it becomes final, and the waves to its right apply the eight steps in a
burst after ONE barrier.  The claim the GPU test pins on the device (tests/test_gpu_round4.py::test_ldl_solver_equals_gauss_jordan: every
bit of dx) is that both are the SAME arithmetic element for element; this file pins the index logic of that claim without a GPU: who
publishes what when, which lanes of which register are the pivot and the column entries, that no cell is read before it is written
(unwritten cells are NaN here) -- in any fixed arithmetic the two schedules must agree bit for bit, and with numpy.linalg.solve to round-off.
"""
import numpy as np
import pytest

NS, NS1, LDA = 71, 72, 73
LANE = np.arange(64)


def _system(n, seed):
    rng = np.random.default_rng(seed)
    J = rng.standard_normal((300, n))
    H = J.T @ J / 300 + np.eye(n) * 0.1
    b = rng.standard_normal(n)
    A = np.full((NS1, LDA), np.nan)      # LDS as the assembly leaves it: [H | b], b also as row n; everything else unwritten
    A[:n, :n] = H
    A[:n, n] = b
    A[n, :n] = b
    return A, H, b


def _load(A, w):
    v0 = np.stack([A[LANE, 8 * w + jj] for jj in range(8)])
    vx = A[64 + (LANE & 7), 8 * w + (LANE >> 3)].copy()
    return v0, vx


def _extract(regs, rdv, n):
    v0, vx = regs[n >> 3]
    bn = v0[n & 7] if n == NS else v0[6]
    dx = np.zeros(n)
    for l in range(64):
        if l < n:
            dx[l] = bn[l] * rdv[l]
        if (l >> 3) == (n & 7) and 64 + (l & 7) < n:
            dx[64 + (l & 7)] = vx[l] * rdv[64 + (l & 7)]
    return dx


def _step(v0, vx, ci0, cix, cjx, rdk, k, w, first_col):
    l0 = np.where(LANE == k, 0.0, ci0 * rdk)
    lx = np.where(64 + (LANE & 7) == k, 0.0, cix * rdk)
    csrc, cbase = (cix, 0) if w == 8 else (ci0, 8 * w)
    for jj in range(first_col, 8):
        v0[jj] = v0[jj] - l0 * csrc[cbase + jj]
    return v0, vx - lx * cjx


def per_pivot(A, n):
    """SOLVER 2: barrier, every wave with an open column reads column k from LDS, the owner of column k + 1 publishes it."""
    A = A.copy()
    regs = {w: _load(A, w) for w in range(9)}
    rdv = np.zeros(NS1)
    for k in range(n):
        new = A.copy()
        for w in range(9):
            if not 8 * w + 7 > k:
                continue
            v0, vx = regs[w]
            ci0, cix, cjx = A[LANE, k], A[64 + (LANE & 7), k], A[8 * w + (LANE >> 3), k]
            rdv[k] = 1.0 / (cix if k >= 64 else ci0)[k & 63]
            v0, vx = _step(v0, vx, ci0, cix, cjx, rdv[k], k, w, 0)
            regs[w] = (v0, vx)
            kb, t = divmod(k, 8)
            tn = (t + 1) & 7
            if w == (kb + 1 if t == 7 else kb):
                new[LANE, k + 1] = v0[tn]
                m = (LANE >> 3) == tn
                new[64 + (LANE & 7)[m], k + 1] = vx[m]
        A = new
    return _extract(regs, rdv, n)


def per_panel(A, n):
    """SOLVER 3: wave kb factorises its panel alone (pivot and column entries are lanes of its own register; rows 64 .. 71 through its own
    LDS write), publishes columns and reciprocals; ONE barrier; the waves to its right apply the eight steps in a burst."""
    A = A.copy()
    rdv = np.full(NS1, np.nan)
    workers = [w for w in range(9) if 8 * w <= n]
    regs = {w: _load(A, w) for w in workers}
    kb = 0
    while 8 * kb < n:
        v0, vx = regs[kb]
        for t in range(8):
            k = 8 * kb + t
            if k >= n:
                continue
            A[LANE, k] = v0[t]
            m = (LANE >> 3) == t
            A[64 + (LANE & 7)[m], k] = vx[m]
            rdv[k] = 1.0 / (v0[t][k] if kb < 8 else vx[9 * t])
            if t < 7 or kb == 8:
                v0, vx = _step(v0, vx, v0[t].copy(), A[64 + (LANE & 7), k], A[8 * kb + (LANE >> 3), k], rdv[k], k, kb, t + 1)
        regs[kb] = (v0, vx)
        for w in workers:                                   # after the barrier
            if w <= kb:
                continue
            v0, vx = regs[w]
            for t in range(8):
                k = 8 * kb + t
                assert k < n                                # the burst needs no guard: 8 kb + 7 < 8 w <= n
                v0, vx = _step(v0, vx, A[LANE, k], A[64 + (LANE & 7), k], A[8 * w + (LANE >> 3), k], rdv[k], k, w, 0)
            regs[w] = (v0, vx)
        kb += 1
    return _extract(regs, rdv, n)


@pytest.mark.parametrize("n", [NS, 6])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_panel_schedule_is_the_per_pivot_schedule(n, seed):
    A, H, b = _system(n, seed)
    with np.errstate(invalid="ignore"):                     # the lanes of unwritten rows carry NaN, as uninitialised LDS may: never into a result
        a, c = per_pivot(A, n), per_panel(A, n)
    x = np.linalg.solve(H, b)
    assert np.isfinite(a).all() and np.array_equal(a, c)
    assert np.abs(c - x).max() <= 1e-12 * max(1.0, np.abs(x).max())
