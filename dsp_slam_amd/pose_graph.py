"""Camera-object pose-graph edge: the consumer of the poses the optimiser produces (include/dsp_pose_graph.h).

Mirrors `EdgeSE3LieAlgebra` / `VertexSE3Object` (reference include/ObjectPoseGraph.h:32-89) and the way the local / global bundle
adjustment sets the edges up (src/Optimizer_util.cc:82-84,190-223,448-450,548-577): one edge per (key frame, object) observation with
the detection's `SE3Tco` as measurement, information `1e3 I`, Huber kernel.  g2o's solver is out of scope; this is the per-edge math
a solver calls, batched over edges, in the C-ABI library (host fp64 -- nothing here is data-parallel enough for the GPU).
Poses are g2o `SE3Quat::toVector()` 7-vectors `[t, qx, qy, qz, qw]`; `from_matrix` is `Converter::toSE3Quat`.
"""
import numpy as np

from . import _lib as L

INV_SIGMA_OBJECT = 1e3                                      # src/Optimizer_util.cc:82,448
# `const float th = sqrt(...)`: the reference keeps the thresholds as float32 (the sqrt itself runs in double on the double-promoted
# product 0.10 * invSigmaObject with a float invSigmaObject, then narrows)
TH_HUBER_OBJECT_JOINT_BA = float(np.float32(np.sqrt(0.10 * float(np.float32(1e3)))))       # :83       Optimizer::JointBundleAdjustment (global)
TH_HUBER_OBJECT_LOCAL_BA = float(np.float32(np.sqrt(float(np.float32(1e3)))))              # :449-450  Optimizer::LocalJointBundleAdjustment
VERTEX_EXPMAP, VERTEX_OBJECT = 0, 1


def _f64(a, width):
    a = np.ascontiguousarray(a, dtype=np.float64)
    single = a.ndim == (2 if width == 16 else 1)
    a = a.reshape(-1, width)
    return a, single


def _call(name, n, *args):
    L.check(getattr(L.load(), name)(n, *args), None, name)


def _p(a):
    return L.ptr(a, L.c_f64p)


def from_matrix(t44):
    """(n,4,4) or (4,4) homogeneous matrices -> (n,7) / (7,) SE3Quat vectors."""
    m, single = _f64(t44, 16)
    out = np.empty((m.shape[0], 7))
    _call("dsp_pg_from_matrix", m.shape[0], _p(m), _p(out))
    return out[0] if single else out


def to_matrix(se3):
    s, single = _f64(se3, 7)
    out = np.empty((s.shape[0], 16))
    _call("dsp_pg_to_matrix", s.shape[0], _p(s), _p(out))
    out = out.reshape(-1, 4, 4)
    return out[0] if single else out


def log(se3):
    s, single = _f64(se3, 7)
    out = np.empty((s.shape[0], 6))
    _call("dsp_pg_log", s.shape[0], _p(s), _p(out))
    return out[0] if single else out


def exp(v6):
    v, single = _f64(v6, 6)
    out = np.empty((v.shape[0], 7))
    _call("dsp_pg_exp", v.shape[0], _p(v), _p(out))
    return out[0] if single else out


def edge_error(v1, v2, meas):
    """computeError: log(meas^-1 * v1 * v2^-1); v1 = Tcw (key frame), v2 = Tow (object), meas = SE3Tco."""
    a, single = _f64(v1, 7)
    b, _ = _f64(v2, 7)
    z, _ = _f64(meas, 7)
    if not (a.shape == b.shape == z.shape):
        raise ValueError("v1, v2 and meas must have the same number of poses")
    out = np.empty((a.shape[0], 6))
    _call("dsp_pg_edge_error", a.shape[0], _p(a), _p(b), _p(z), _p(out))
    return out[0] if single else out


def edge_linearize(meas, err):
    """linearizeOplus: (j_xi, j_xj), each (n,6,6) row-major."""
    z, single = _f64(meas, 7)
    e, _ = _f64(err, 6)
    if z.shape[0] != e.shape[0]:
        raise ValueError("meas and err must have the same number of edges")
    ji = np.empty((z.shape[0], 6, 6))
    jj = np.empty((z.shape[0], 6, 6))
    _call("dsp_pg_edge_linearize", z.shape[0], _p(z), _p(e), _p(ji), _p(jj))
    return (ji[0], jj[0]) if single else (ji, jj)


def edge_chi2(err, inv_sigma=INV_SIGMA_OBJECT, huber_delta=0.0):
    """(chi2, rho, weight): chi2 = err^T (inv_sigma I) err; rho / weight from g2o's RobustKernelHuber (delta <= 0: none)."""
    e, single = _f64(err, 6)
    chi2, rho, w = np.empty(e.shape[0]), np.empty(e.shape[0]), np.empty(e.shape[0])
    _call("dsp_pg_edge_chi2", e.shape[0], _p(e), float(inv_sigma), float(huber_delta), _p(chi2), _p(rho), _p(w))
    return (chi2[0], rho[0], w[0]) if single else (chi2, rho, w)


def vertex_oplus(estimate, update, kind=VERTEX_EXPMAP):
    s, single = _f64(estimate, 7)
    u, _ = _f64(update, 6)
    if s.shape[0] != u.shape[0]:
        raise ValueError("estimate and update must have the same number of vertices")
    out = np.empty((s.shape[0], 7))
    L.check(L.load().dsp_pg_vertex_oplus(s.shape[0], int(kind), _p(s), _p(u), _p(out)), None, "dsp_pg_vertex_oplus")
    return out[0] if single else out
