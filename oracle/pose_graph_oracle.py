"""TEST INFRASTRUCTURE ONLY (never imported by dsp_slam_amd/): numpy restatement of the camera-object pose-graph edge that
consumes the `SE3Tco` measurements the pose optimiser produces.

Follows, function by function (fp64 like g2o):
  SE3Quat (7-vector [t, qx, qy, qz, qw])         Thirdparty/g2o/g2o/types/se3quat.h:45-300
    ctor normalisation :286-291, operator* :105-111, inverse :123-128, log :178-217, exp :225-262, adj :264-273,
    6-vector ctor [t | q.xyz] :70-84
  skew, deltaR                                    Thirdparty/g2o/g2o/types/se3_ops.hpp:27-49
  EdgeSE3LieAlgebra::computeError / linearizeOplus include/ObjectPoseGraph.h:70-88
  VertexSE3Object::oplusImpl                      include/ObjectPoseGraph.h:50-54
  VertexSE3Expmap::oplusImpl                      Thirdparty/g2o/g2o/types/types_six_dof_expmap.h:73-76 (exp(update) * estimate)
  chi2 / RobustKernelHuber                        Thirdparty/g2o/g2o/core/base_edge.h:58-61, robust_kernel_impl.cpp:78-91
  edge set-up (information 1e3 I, Huber delta)    src/Optimizer_util.cc:82-84,210-223,448-450,566-577
  Eigen::Quaterniond <-> Matrix3d                 Eigen/src/Geometry/Quaternion.h (toRotationMatrix, quaternion from matrix),
                                                  third-party, absent from /root/reference: restated from its published algorithm.

PARITY UNPINNED: g2o needs Eigen, which is not in this image, so the reference's edge cannot be compiled here and the
reference holds no test or golden vector for it.  What pins this file are the identities the tests check (log(exp x) = x,
error 0 at the measurement, jacobians against finite differences to the order the reference's own approximation has).
"""
import numpy as np


def skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def delta_r(R):
    return np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])


# ---- Eigen quaternion helpers (coefficients stored x, y, z, w) -------------------------------------------------------------
def quat_to_rot(q):
    x, y, z, w = q
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1.0 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1.0 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1.0 - (txx + tyy)]])


def rot_to_quat(m):
    q = np.zeros(4)
    t = m[0, 0] + m[1, 1] + m[2, 2]
    if t > 0.0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (m[2, 1] - m[1, 2]) * t
        q[1] = (m[0, 2] - m[2, 0]) * t
        q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    return q


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def quat_rotate(q, v):
    uv = 2.0 * np.cross(q[:3], v)
    return v + q[3] * uv + np.cross(q[:3], uv)


def normalize_rotation(q):
    q = np.array(q, np.float64)
    if q[3] < 0:
        q = -q
    return q / np.sqrt(np.dot(q, q))


# ---- SE3Quat as a 7-vector [t, qx, qy, qz, qw] (toVector order) -------------------------------------------------------------
def se3(q, t):
    return np.concatenate([np.asarray(t, np.float64), normalize_rotation(q)])


def se3_from_matrix(T):
    """Converter::toSE3Quat: SE3Quat(R, t) -> quaternion from the matrix, then normalizeRotation."""
    T = np.asarray(T, np.float64)
    return se3(rot_to_quat(T[:3, :3]), T[:3, 3])


def se3_to_matrix(s):
    T = np.eye(4)
    T[:3, :3] = quat_to_rot(s[3:])
    T[:3, 3] = s[:3]
    return T


def se3_from_minimal(v):
    """The templated 6-vector constructor (se3quat.h:70-84): [t | q.xyz], w recovered -- NOT the exponential map."""
    v = np.asarray(v, np.float64)
    q = np.array([v[3], v[4], v[5], 0.0])
    n = np.sqrt(np.dot(q, q))
    if n > 1.0:
        q = q / n
    else:
        w2 = 1.0 - np.dot(q, q)
        q[3] = 0.0 if w2 < 0.0 else np.sqrt(w2)
    return np.concatenate([v[:3], q])


def se3_mul(a, b):
    t = a[:3] + quat_rotate(a[3:], b[:3])
    return np.concatenate([t, normalize_rotation(quat_mul(a[3:], b[3:]))])


def se3_inverse(a):
    qc = np.array([-a[3], -a[4], -a[5], a[6]])
    return np.concatenate([quat_rotate(qc, -a[:3]), qc])


def se3_log(a):
    R = quat_to_rot(a[3:])
    d = 0.5 * (R[0, 0] + R[1, 1] + R[2, 2] - 1.0)
    dR = delta_r(R)
    if d > 0.99999:
        omega = 0.5 * dR
        Om = skew(omega)
        v_inv = np.eye(3) - 0.5 * Om + (1.0 / 12.0) * (Om @ Om)
    else:
        theta = np.arccos(d)
        omega = theta / (2.0 * np.sqrt(1.0 - d * d)) * dR
        Om = skew(omega)
        v_inv = np.eye(3) - 0.5 * Om + (1.0 - theta / (2.0 * np.tan(theta / 2.0))) / (theta * theta) * (Om @ Om)
    return np.concatenate([omega, v_inv @ a[:3]])


def se3_exp(u):
    u = np.asarray(u, np.float64)
    omega, ups = u[:3], u[3:]
    theta = np.sqrt(np.dot(omega, omega))
    Om = skew(omega)
    if theta < 0.00001:
        R = np.eye(3) + Om + Om @ Om
        V = R
    else:
        Om2 = Om @ Om
        R = np.eye(3) + np.sin(theta) / theta * Om + (1.0 - np.cos(theta)) / (theta * theta) * Om2
        V = np.eye(3) + (1.0 - np.cos(theta)) / (theta * theta) * Om + (theta - np.sin(theta)) / (theta ** 3) * Om2
    return se3(rot_to_quat(R), V @ ups)


def se3_adj(a):
    R = quat_to_rot(a[3:])
    res = np.zeros((6, 6))
    res[:3, :3] = R
    res[3:, 3:] = R
    res[3:, :3] = skew(a[:3]) @ R
    return res


# ---- the edge and the vertices -----------------------------------------------------------------------------------------------
def edge_error(v1, v2, meas):
    """_error = (Z^-1 * Ti * Tj^-1).log()   (ObjectPoseGraph.h:70-74); Ti = Tcw, Tj = Tow, Z = Tco."""
    return se3_log(se3_mul(se3_mul(se3_inverse(meas), v1), se3_inverse(v2)))


def edge_linearize(meas, err):
    """ObjectPoseGraph.h:76-88: J = I + 0.5 [[w]x 0; [t]x [w]x];  dXi = J Ad(Z^-1), dXj = -J."""
    w, t = err[:3], err[3:]
    J = np.zeros((6, 6))
    J[:3, :3] = skew(w)
    J[3:, :3] = skew(t)
    J[3:, 3:] = skew(w)
    J = 0.5 * J + np.eye(6)
    return J @ se3_adj(se3_inverse(meas)), -J


def vertex_oplus_expmap(est, update):
    return se3_mul(se3_exp(update), est)


def vertex_oplus_object(est, update):
    """VertexSE3Object::oplusImpl: `SE3Quat s(update)` is the 6-vector constructor, then estimate * s^-1."""
    return se3_mul(est, se3_inverse(se3_from_minimal(update)))


def edge_chi2(err, inv_sigma, huber_delta):
    """chi2 = e^T (inv_sigma I) e; Huber: rho = e2 (e2 <= d^2) else 2 d sqrt(e2) - d^2; weight rho' = 1 or d / sqrt(e2)."""
    e2 = inv_sigma * float(np.dot(err, err))
    d2 = huber_delta * huber_delta
    if huber_delta <= 0 or e2 <= d2:
        return e2, e2, 1.0
    s = np.sqrt(e2)
    return e2, 2.0 * s * huber_delta - d2, huber_delta / s
