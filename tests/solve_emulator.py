"""CPU emulation of k_solve<0> (gn_kernels.hip): the LDL^T factorisation of the augmented normal equations with the lower triangle held
as packed, column-major elements in the registers of 512 threads, one LDS publication per element, and the back substitution by one
wave.  Test infrastructure: it follows the kernel's data flow statement by statement (same packing formula, same publication rule, same
lane assignment in the back substitution), so that the algorithm -- index mapping, which step may read what, the b-as-row-n trick -- is
checked on the CPU (tests/test_solve_emulation.py); the GPU tests then compare the kernel itself with the Gauss-Jordan kernel and the
oracle."""
import numpy as np

NSOLVE = 71
NS1 = NSOLVE + 1
LDL_THREADS = 512
LDL_NP = NS1 * (NS1 + 1) // 2
LDL_EPT = (LDL_NP + LDL_THREADS - 1) // LDL_THREADS


def packed_to_ij(e):
    """The kernel's closed form + fix-up loops: packed column-major index of the 72 x 72 lower triangle -> (i, j)."""
    j = int((2 * NS1 + 1 - np.sqrt(np.float32((2 * NS1 + 1) * (2 * NS1 + 1) - 8 * e))) * np.float32(0.5))
    j = min(max(j, 0), NS1 - 1)
    while j > 0 and j * NS1 - j * (j - 1) // 2 > e:
        j -= 1
    while (j + 1) * NS1 - (j + 1) * j // 2 <= e:
        j += 1
    return j + (e - (j * NS1 - j * (j - 1) // 2)), j


def fast_recip(d):
    r = 1.0 / np.float32(d)           # stands in for v_rcp_f64's ~25 good bits
    r = float(r)
    r = r + r * (1.0 - d * r)
    r = r + r * (1.0 - d * r)
    return r


def ldl_solve(H, b):
    """H (n, n) symmetric positive definite, b (n,), n <= 71 -> (dx, singular)."""
    n = H.shape[0]
    A = np.zeros((NS1, NS1 + 1))
    A[:n, :n] = H
    A[:n, n] = b
    A[n, :n] = A[:n, n]                                  # b as row n
    rdv = np.zeros(NS1)
    rdv[0] = fast_recip(A[0, 0])
    sing = not (A[0, 0] > 0.0)
    # registers: thread t holds elements e = t + 256 q
    ei = np.zeros((LDL_THREADS, LDL_EPT), int)
    ej = -np.ones((LDL_THREADS, LDL_EPT), int)
    v = np.zeros((LDL_THREADS, LDL_EPT))
    for t in range(LDL_THREADS):
        for q in range(LDL_EPT):
            e = t + LDL_THREADS * q
            if e < LDL_NP:
                i, j = packed_to_ij(e)
                if i <= n and j <= n and not (i == n and j == n):
                    ei[t, q], ej[t, q], v[t, q] = i, j, A[i, j]
    for k in range(n):
        # barrier: column k and rdv[k] are visible; writes of this step go to column k + 1 only
        col_k = A[:, k].copy()
        rdk = rdv[k]
        for t in range(LDL_THREADS):
            for q in range(LDL_EPT):
                if ej[t, q] > k:
                    cik, cjk = col_k[ei[t, q]], col_k[ej[t, q]]
                    v[t, q] = v[t, q] - (cik * rdk) * cjk
                    if ej[t, q] == k + 1:
                        A[ei[t, q], k + 1] = v[t, q]
                        if ei[t, q] == k + 1:
                            rdv[k + 1] = fast_recip(v[t, q])
                            if not (v[t, q] > 0.0):
                                sing = True
    if sing:
        return None, True
    # back substitution by 64 lanes
    w = np.zeros((64, 2))
    rd = np.zeros((64, 2))
    for lane in range(64):
        for s, j in enumerate((lane, lane + 64)):
            if j < n:
                w[lane, s], rd[lane, s] = A[n, j], rdv[j]
    dx = np.zeros(n)
    for i in range(n - 1, -1, -1):
        src = i & 63
        xi = rd[src, 0] * w[src, 0] if i < 64 else rd[src, 1] * w[src, 1]
        dx[i] = xi
        for lane in range(64):
            for s, j in enumerate((lane, lane + 64)):
                cc = A[i, j] if j < i else 0.0
                w[lane, s] = w[lane, s] - cc * xi
    return dx, False


def solve_rows_in_lanes(H, b):
    """k_solve<2> (default): the factorisation with rows in lanes.  Wave w < 9 owns columns 8w .. 8w+7; lane l holds v0[jj] =
    A[l][8w+jj] (rows 0 .. 63) and vx = A[64 + (l & 7)][8w + (l >> 3)] (rows 64 .. 71).  Per step a wave reads its rows' entries of column
    k (ci0, cix) and vx's column entry (cjx) from LDS; the pivot and the eight column entries c_jk are other lanes' values of ci0 / cix
    (v_readlane).  Every row except the pivot row is eliminated (Gauss-Jordan on the symmetric trailing part), so column n ends as
    d_i dx_i and there is no back substitution.  Waves whose columns are all <= k skip the step.  Column k + 1 is published by its owner
    for all rows.  Rows and columns beyond the system (n = 6: everything past index 6) hold garbage -- NaN here -- that must stay where
    it is."""
    n = H.shape[0]
    A = np.full((NS1, NS1 + 1), np.nan)
    A[:n, :n] = H
    A[:n, n] = b
    A[n, :n] = A[:n, n]
    rdv = np.full(NS1, np.nan)
    lanes = np.arange(64)
    xr, xj = 64 + (lanes & 7), lanes >> 3
    v0 = [np.array([[A[l, 8 * w + jj] for jj in range(8)] for l in lanes]) for w in range(9)]          # [w][lane, jj]
    vx = [np.array([A[xr[l], 8 * w + xj[l]] for l in lanes]) for w in range(9)]
    sing = False
    with np.errstate(invalid="ignore"):
        for k in range(n):
            col = A[:, k].copy()                              # barrier: what the waves read this step
            for w in range(9):
                if not 8 * w + 7 > k:
                    continue
                ci0, cix, cjx = col[lanes], col[xr], col[8 * w + xj]
                d = (cix if k >= 64 else ci0)[k & 63]
                rdk = fast_recip(d)
                if w == 8:
                    sing = sing or not (d > 0.0)
                    rdv[k] = rdk
                csrc, cbase = (cix, 0) if w == 8 else (ci0, 8 * w)
                l0 = np.where(lanes == k, 0.0, ci0 * rdk)
                lx = np.where(xr == k, 0.0, cix * rdk)
                for jj in range(8):
                    v0[w][:, jj] = v0[w][:, jj] - l0 * csrc[cbase + jj]
                vx[w] = vx[w] - lx * cjx
                if w == (k + 1) >> 3:
                    tn = (k + 1) & 7
                    A[lanes, k + 1] = v0[w][:, tn]
                    sel = xj == tn
                    A[xr[sel], k + 1] = vx[w][sel]
    if sing:
        return None, True
    wn, jn = n >> 3, n & 7
    dx = np.zeros(n)
    for i in range(min(n, 64)):
        dx[i] = v0[wn][i, jn] * rdv[i]
    for l in lanes:
        if xj[l] == jn and xr[l] < n:
            dx[xr[l]] = vx[wn][l] * rdv[xr[l]]
    return dx, False
