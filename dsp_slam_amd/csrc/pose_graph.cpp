// Camera-object pose-graph edge arithmetic (include/dsp_pose_graph.h): host fp64, g2o conventions.
//
// Replaces the per-edge math of   include/ObjectPoseGraph.h:50-54,70-88   and the g2o::SE3Quat operations it calls
// (Thirdparty/g2o/g2o/types/se3quat.h:105-111,123-128,178-273, se3_ops.hpp:27-49); the quaternion <-> matrix conversions are Eigen's
// published algorithms (Eigen/src/Geometry/Quaternion.h), written out because Eigen is a dependency the drop-in does not take.
// No GPU work: SURVEY.md section 8 f4 ("sparse g2o problem -- not data-parallel").  Compiled with -ffp-contract=off.
#include <cmath>
#include <cstdint>

#include "dsp_gn.h"
#include "dsp_pose_graph.h"

namespace {

struct V3 { double x, y, z; };
struct Q4 { double x, y, z, w; };
struct M3 { double m[3][3]; };
struct Pose { V3 t; Q4 q; };

inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

M3 skew(V3 v) { return {{{0.0, -v.z, v.y}, {v.z, 0.0, -v.x}, {-v.y, v.x, 0.0}}}; }
M3 identity() { return {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}}; }
M3 mul(const M3& a, const M3& b) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}
V3 mul(const M3& a, V3 v) {
    return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
            a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
// a I + b A + c B
M3 combine(double a, double b, const M3& A, double c, const M3& B) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = (i == j ? a : 0.0) + b * A.m[i][j] + c * B.m[i][j];
    return r;
}

M3 to_rotation(Q4 q) {   // Eigen QuaternionBase::toRotationMatrix
    const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    return {{{1.0 - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1.0 - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1.0 - (txx + tyy)}}};
}

Q4 from_rotation(const M3& r) {   // Eigen quaternionbase_assign_impl<Matrix3>
    double q[4];
    double t = r.m[0][0] + r.m[1][1] + r.m[2][2];
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (r.m[2][1] - r.m[1][2]) * t;
        q[1] = (r.m[0][2] - r.m[2][0]) * t;
        q[2] = (r.m[1][0] - r.m[0][1]) * t;
    } else {
        int i = 0;
        if (r.m[1][1] > r.m[0][0]) i = 1;
        if (r.m[2][2] > r.m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(r.m[i][i] - r.m[j][j] - r.m[k][k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (r.m[k][j] - r.m[j][k]) * t;
        q[j] = (r.m[j][i] + r.m[i][j]) * t;
        q[k] = (r.m[k][i] + r.m[i][k]) * t;
    }
    return {q[0], q[1], q[2], q[3]};
}

Q4 qmul(Q4 a, Q4 b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
V3 rotate(Q4 q, V3 v) {   // Eigen QuaternionBase::_transformVector
    const V3 u = {q.x, q.y, q.z};
    const V3 uv = 2.0 * cross(u, v);
    return v + (q.w * uv) + cross(u, uv);
}
Q4 normalize_rotation(Q4 q) {   // se3quat.h:286-291
    if (q.w < 0) q = {-q.x, -q.y, -q.z, -q.w};
    const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return {q.x / n, q.y / n, q.z / n, q.w / n};
}

// the 7-vector constructor of SE3Quat normalises the rotation (sign flipped to w >= 0, unit norm): se3quat.h:86-92
Pose load(const double* p) { return {{p[0], p[1], p[2]}, normalize_rotation({p[3], p[4], p[5], p[6]})}; }
void store(const Pose& s, double* p) {
    p[0] = s.t.x; p[1] = s.t.y; p[2] = s.t.z;
    p[3] = s.q.x; p[4] = s.q.y; p[5] = s.q.z; p[6] = s.q.w;
}
Pose compose(const Pose& a, const Pose& b) { return {a.t + rotate(a.q, b.t), normalize_rotation(qmul(a.q, b.q))}; }   // se3quat.h:105-111
Pose inverse(const Pose& a) {                                                                                         // se3quat.h:123-128
    const Q4 c = {-a.q.x, -a.q.y, -a.q.z, a.q.w};
    return {rotate(c, -1.0 * a.t), c};
}

void se3_log(const Pose& a, double* out) {   // se3quat.h:178-217
    const M3 R = to_rotation(a.q);
    const double d = 0.5 * (R.m[0][0] + R.m[1][1] + R.m[2][2] - 1.0);
    const V3 dR = {R.m[2][1] - R.m[1][2], R.m[0][2] - R.m[2][0], R.m[1][0] - R.m[0][1]};
    V3 omega;
    M3 v_inv;
    if (d > 0.99999) {
        omega = 0.5 * dR;
        const M3 Om = skew(omega);
        v_inv = combine(1.0, -0.5, Om, 1.0 / 12.0, mul(Om, Om));
    } else {
        const double theta = std::acos(d);
        omega = (theta / (2.0 * std::sqrt(1.0 - d * d))) * dR;
        const M3 Om = skew(omega);
        v_inv = combine(1.0, -0.5, Om, (1.0 - theta / (2.0 * std::tan(theta / 2.0))) / (theta * theta), mul(Om, Om));
    }
    const V3 ups = mul(v_inv, a.t);
    out[0] = omega.x; out[1] = omega.y; out[2] = omega.z;
    out[3] = ups.x; out[4] = ups.y; out[5] = ups.z;
}

Pose se3_exp(const double* u) {   // se3quat.h:225-262
    const V3 omega = {u[0], u[1], u[2]}, ups = {u[3], u[4], u[5]};
    const double theta = std::sqrt(omega.x * omega.x + omega.y * omega.y + omega.z * omega.z);
    const M3 Om = skew(omega);
    const M3 Om2 = mul(Om, Om);
    M3 R, V;
    if (theta < 0.00001) {
        R = combine(1.0, 1.0, Om, 1.0, Om2);
        V = R;
    } else {
        R = combine(1.0, std::sin(theta) / theta, Om, (1.0 - std::cos(theta)) / (theta * theta), Om2);
        V = combine(1.0, (1.0 - std::cos(theta)) / (theta * theta), Om, (theta - std::sin(theta)) / std::pow(theta, 3), Om2);
    }
    return {mul(V, ups), normalize_rotation(from_rotation(R))};
}

void adjoint(const Pose& a, double A[6][6]) {   // se3quat.h:264-273
    const M3 R = to_rotation(a.q);
    const M3 tR = mul(skew(a.t), R);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[i][j] = R.m[i][j];
            A[i + 3][j + 3] = R.m[i][j];
            A[i + 3][j] = tR.m[i][j];
            A[i][j + 3] = 0.0;
        }
}

Pose from_minimal(const double* v) {   // the 6-vector constructor, se3quat.h:70-84
    Q4 q = {v[3], v[4], v[5], 0.0};
    const double n2 = q.x * q.x + q.y * q.y + q.z * q.z;
    if (std::sqrt(n2) > 1.0) {
        const double n = std::sqrt(n2);
        q = {q.x / n, q.y / n, q.z / n, 0.0};
    } else {
        const double w2 = 1.0 - n2;
        q.w = (w2 < 0.0) ? 0.0 : std::sqrt(w2);
    }
    return {{v[0], v[1], v[2]}, q};
}

inline bool bad(int64_t n, const void* a, const void* b) { return n < 0 || (n > 0 && (!a || !b)); }

}  // namespace

extern "C" {

int dsp_pg_from_matrix(int64_t n, const double* t44, double* out) {
    if (bad(n, t44, out)) return DSP_E_ARG;
    for (int64_t e = 0; e < n; ++e) {
        const double* T = t44 + 16 * e;
        const M3 R = {{{T[0], T[1], T[2]}, {T[4], T[5], T[6]}, {T[8], T[9], T[10]}}};
        store({{T[3], T[7], T[11]}, normalize_rotation(from_rotation(R))}, out + 7 * e);
    }
    return DSP_OK;
}

int dsp_pg_to_matrix(int64_t n, const double* se3, double* out) {
    if (bad(n, se3, out)) return DSP_E_ARG;
    for (int64_t e = 0; e < n; ++e) {
        const Pose p = load(se3 + 7 * e);
        const M3 R = to_rotation(p.q);
        double* T = out + 16 * e;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) T[4 * i + j] = R.m[i][j];
            T[12 + i] = 0.0;
        }
        T[3] = p.t.x; T[7] = p.t.y; T[11] = p.t.z; T[15] = 1.0;
    }
    return DSP_OK;
}

int dsp_pg_log(int64_t n, const double* se3, double* out) {
    if (bad(n, se3, out)) return DSP_E_ARG;
    for (int64_t e = 0; e < n; ++e) se3_log(load(se3 + 7 * e), out + 6 * e);
    return DSP_OK;
}

int dsp_pg_exp(int64_t n, const double* v6, double* out) {
    if (bad(n, v6, out)) return DSP_E_ARG;
    for (int64_t e = 0; e < n; ++e) store(se3_exp(v6 + 6 * e), out + 7 * e);
    return DSP_OK;
}

int dsp_pg_edge_error(int64_t n, const double* v1, const double* v2, const double* meas, double* err) {
    if (bad(n, v1, v2) || bad(n, meas, err)) return DSP_E_ARG;
    for (int64_t e = 0; e < n; ++e)   // (Z^-1 * Ti) * Tj^-1, left to right like the C++ expression
        se3_log(compose(compose(inverse(load(meas + 7 * e)), load(v1 + 7 * e)), inverse(load(v2 + 7 * e))), err + 6 * e);
    return DSP_OK;
}

int dsp_pg_edge_linearize(int64_t n, const double* meas, const double* err, double* j_xi, double* j_xj) {
    if (bad(n, meas, err) || bad(n, j_xi, j_xj)) return DSP_E_ARG;
    for (int64_t e = 0; e < n; ++e) {
        const double* er = err + 6 * e;
        const M3 W = skew({er[0], er[1], er[2]}), T = skew({er[3], er[4], er[5]});
        double J[6][6], A[6][6];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                J[i][j] = 0.5 * W.m[i][j] + (i == j ? 1.0 : 0.0);
                J[i][j + 3] = 0.0;
                J[i + 3][j] = 0.5 * T.m[i][j];
                J[i + 3][j + 3] = 0.5 * W.m[i][j] + (i == j ? 1.0 : 0.0);
            }
        adjoint(inverse(load(meas + 7 * e)), A);
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
                double s = 0.0;
                for (int k = 0; k < 6; ++k) s += J[i][k] * A[k][j];
                j_xi[36 * e + 6 * i + j] = s;
                j_xj[36 * e + 6 * i + j] = -J[i][j];
            }
    }
    return DSP_OK;
}

int dsp_pg_edge_chi2(int64_t n, const double* err, double inv_sigma, double huber_delta, double* chi2, double* rho, double* weight) {
    if (bad(n, err, chi2)) return DSP_E_ARG;
    const double d2 = huber_delta * huber_delta;
    for (int64_t e = 0; e < n; ++e) {
        const double* er = err + 6 * e;
        double dot = 0.0;
        for (int k = 0; k < 6; ++k) dot += er[k] * er[k];
        const double e2 = inv_sigma * dot;
        chi2[e] = e2;
        double r = e2, w = 1.0;
        if (huber_delta > 0.0 && !(e2 <= d2)) {   // robust_kernel_impl.cpp:78-91
            const double s = std::sqrt(e2);
            r = 2.0 * s * huber_delta - d2;
            w = huber_delta / s;
        }
        if (rho) rho[e] = r;
        if (weight) weight[e] = w;
    }
    return DSP_OK;
}

int dsp_pg_vertex_oplus(int64_t n, int kind, const double* estimate, const double* update, double* out) {
    if (bad(n, estimate, update) || (n > 0 && !out) || (kind != DSP_PG_VERTEX_EXPMAP && kind != DSP_PG_VERTEX_OBJECT)) return DSP_E_ARG;
    for (int64_t e = 0; e < n; ++e) {
        const Pose est = load(estimate + 7 * e);
        store(kind == DSP_PG_VERTEX_EXPMAP ? compose(se3_exp(update + 6 * e), est) : compose(est, inverse(from_minimal(update + 6 * e))), out + 7 * e);
    }
    return DSP_OK;
}

}  // extern "C"
