/* dsp_pose_graph.h -- the camera-object pose-graph edge that consumes the optimiser's SE3Tco measurements (C ABI, host fp64).
 *
 * DSP-SLAM feeds every pose `Optimizer.estimate_pose_cam_obj` / `reconstruct_object` returns into its local and global
 * bundle adjustment as the measurement of a 6-dof edge between the key-frame vertex (Tcw) and the object vertex (Tow):
 *   EdgeSE3LieAlgebra::computeError / linearizeOplus    include/ObjectPoseGraph.h:70-88
 *   VertexSE3Object::oplusImpl                           include/ObjectPoseGraph.h:50-54
 *   VertexSE3Expmap::oplusImpl (the vertex type the optimiser actually instantiates: src/Optimizer_util.cc:190-194,548-552)
 *                                                        Thirdparty/g2o/g2o/types/types_six_dof_expmap.h:73-76
 *   edge set-up: information = 1e3 I, Huber delta         src/Optimizer_util.cc:82-84,210-223,448-450,566-577
 *   outlier test chi2 > thHuberObjectSquare               src/Optimizer_util.cc:647-656,701-706
 * g2o's sparse solver itself is out of scope ("not data-parallel"); these entry points are the
 * per-edge arithmetic a g2o edge class (or any other solver) delegates to, in g2o's own conventions:
 *   - a pose is g2o::SE3Quat::toVector(): 7 doubles [tx ty tz qx qy qz qw] (se3quat.h:139-149), rotation normalised with qw >= 0;
 *   - an error / update is 6 doubles [omega(3) | upsilon(3)] (se3quat.h:178-217);
 *   - jacobians are 6x6, ROW-major here (Eigen's default for Matrix6d is column-major: transpose on the way in).
 * Host code, double precision like g2o; no GPU is touched and no dsp_handle is needed.  All functions return DSP_OK or DSP_E_ARG.
 */
#ifndef DSP_POSE_GRAPH_H
#define DSP_POSE_GRAPH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Converter::toSE3Quat (src/Converter.cc): row-major 4x4 [R t; 0 1] -> 7-vector (quaternion from the matrix as Eigen does, then
 * SE3Quat's normalizeRotation), and back (SE3Quat::to_homogeneous_matrix). */
int dsp_pg_from_matrix(int64_t n, const double* t44, double* se3_out);
int dsp_pg_to_matrix(int64_t n, const double* se3, double* t44_out);

/* SE3Quat::log (se3quat.h:178-217) and SE3Quat::exp (:225-262), including their small-angle branches as written. */
int dsp_pg_log(int64_t n, const double* se3, double* v6_out);
int dsp_pg_exp(int64_t n, const double* v6, double* se3_out);

/* EdgeSE3LieAlgebra::computeError: err[e] = log(meas[e]^-1 * v1[e] * v2[e]^-1)   (v1 = Tcw of the key frame, v2 = Tow of the object,
 * meas = SE3Tco of the detection). */
int dsp_pg_edge_error(int64_t n, const double* v1, const double* v2, const double* meas, double* err_out);

/* EdgeSE3LieAlgebra::linearizeOplus from the error computed above: J = I + 0.5 [[w]x 0; [t]x [w]x];
 * j_xi = J * Ad(meas^-1), j_xj = -J   (n x 6 x 6, row-major). */
int dsp_pg_edge_linearize(int64_t n, const double* meas, const double* err, double* j_xi_out, double* j_xj_out);

/* chi2 = err^T (inv_sigma I) err (g2o BaseEdge::chi2) and RobustKernelHuber::robustify (robust_kernel_impl.cpp:78-91):
 * rho = chi2 if chi2 <= delta^2 else 2 delta sqrt(chi2) - delta^2; weight = rho'.  huber_delta <= 0: no kernel (weight 1).
 * rho_out / weight_out may be NULL. */
int dsp_pg_edge_chi2(int64_t n, const double* err, double inv_sigma, double huber_delta, double* chi2_out, double* rho_out,
                     double* weight_out);

/* Vertex update.  kind DSP_PG_VERTEX_EXPMAP: exp(update) * estimate (VertexSE3Expmap);  DSP_PG_VERTEX_OBJECT:
 * estimate * SE3Quat(update)^-1 where SE3Quat(update) is g2o's 6-vector constructor [t | q.xyz] (se3quat.h:70-84), as
 * VertexSE3Object::oplusImpl is written. */
#define DSP_PG_VERTEX_EXPMAP 0
#define DSP_PG_VERTEX_OBJECT 1
int dsp_pg_vertex_oplus(int64_t n, int kind, const double* estimate, const double* update, double* se3_out);

#ifdef __cplusplus
}
#endif
#endif
