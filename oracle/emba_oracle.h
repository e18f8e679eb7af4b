/* oracle/emba_oracle.h — TEST INFRASTRUCTURE. NOT PART OF THE PRODUCT.
 *
 * Plain-C, double-precision, single-threaded CPU restatement of the EMBA hot path
 * (SURVEY.md §8a rows a1-a12), following the reference's evaluation order including
 * its quirks Q1-Q10.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library, and only as the checker.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   a3 (SO(3) linear spline value + Jacobian, Sophus exp/log, basalt Jl/Jl^-1):
 *        PINNED against the reference's own unmodified headers compiled into
 *        oracle/_ref/libref_basalt.so and against tests/golden/so3_spline_n2.npz.
 *   a1,a2,a4-a12 (ros::Time midpoint, cv::Sobel, warp, projection, pairing, normal eq.):
 *        PARITY UNPINNED — the reference holds no golden vectors / assertions for them
 *         and model.cpp / event_pano_warper.cpp / trajectory.cpp /
 *        equirectangular_camera.h need ROS, OpenCV and glog headers that this image
 *        lacks (unbuildable here without stand-ins, which are not allowed).  They are
 *        restated line by line from the cited reference source and checked by
 *        numeric-differentiation and hand-derived known-answer tests only.
 */
#ifndef EMBA_ORACLE_H
#define EMBA_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct emba_oracle emba_oracle;

/* LEGM::LEGM (src/emba/model.cpp:56-70) + EventWarper::initialize
 * (src/utils/event_pano_warper.cpp:7-25).  The bearing LUT (S*3 doubles, row-major by
 * sensor pixel y*w+x) is an INPUT, replacing precomputeBearingVectors (:27-41). */
emba_oracle* emba_oracle_create(int sensor_w, int sensor_h, int pano_w, int pano_h,
                                const double* bearing_lut, double C_th);
void emba_oracle_destroy(emba_oracle* o);

/* Shard view: events [0, k0) of the next evaluateDataError are warped and serve as predecessors but form no
 * measurement themselves — what a time shard's halo does.  k0 = 0 (default) is the reference.  Used by the tests that check one
 * rank's shard of a 40 M / 100 M-event stream without evaluating the whole stream on the CPU. */
void emba_oracle_set_first_counted(emba_oracle* o, size_t k0);

/* a2: t_batch = t_bgn + (t_end - t_bgn) * 0.5 in ros::Time / ros::Duration arithmetic
 * (src/emba/model.cpp:116-119; rostime semantics per SURVEY.md Appendix A). */
int64_t emba_oracle_batch_mid_ns(int64_t t_first_ns, int64_t t_last_ns);

/* a3: LinearTrajectory::evaluate (src/utils/trajectory.cpp:122-147) ->
 * basalt::So3Spline<2>::evaluate (so3_spline.h:218-274).  q_xyzw[4], R[9] row-major,
 * J36[18] row-major 3x6 = [d/dknot_s | d/dknot_{s+1}].  Returns 0 ok, 1 if t outside knots. */
int emba_oracle_spline_eval(const double* knots_xyzw, int K, int64_t t0_ns, int64_t dt_ns,
                            int64_t t_ns, double* q_xyzw, double* R, int* cp_idx, double* J36);

/* Sophus::SO3d::exp / log (so3.hpp:583-619, 247-290); basalt leftJacobianSO3 / InvSO3
 * (sophus_utils.hpp:332-414).  Exposed for branch-level pinning against oracle/_ref. */
void emba_oracle_so3_exp(const double* w, double* q_xyzw);
void emba_oracle_so3_log(const double* q_xyzw, double* w);
void emba_oracle_left_jacobian(const double* phi, double* J, double* Jinv);

/* a5: EquirectangularCamera::projectToImage (include/utils/equirectangular_camera.h:18-45,64-67).
 * J23 row-major 2x3 (may be NULL). */
void emba_oracle_project(int pano_w, int pano_h, const double* rb, double* pm, double* J23);

/* a4: EventWarper::warpEventToMap (src/utils/event_pano_warper.cpp:43-74). */
void emba_oracle_warp(const emba_oracle* o, int ev_x, int ev_y, const double* q_xyzw, double* pm,
                      double* J23);

/* a1: 0.125*Sobel3x3 (BORDER_REFLECT_101), Gxy := (d/dy Gx + d/dx Gy)/2  (model.cpp:88-97). */
void emba_oracle_hessian(const double* Gx, const double* Gy, int H, int W, double* Gxx, double* Gxy,
                         double* Gyy);

/* Optional per-event dump of the state evaluateDataError leaves behind (State_LEGM,
 * include/emba/state.h:56-83), in ORIGINAL (time) event order; any pointer may be NULL.
 *   pm[2n], D[12n] (dpm_ddrot_cp, row-major 2x6), cp_idx[n], inlier_idx[n]
 *   (-1 outlier, -2 not a measurement: first event at its pixel or dropped tail event),
 *   pm_int[2n], dp[2n], Gpm[2n], temp[2n], prev[n] (index of the predecessor event or -1). */
typedef struct {
    double* pm; double* D; int32_t* cp_idx; int32_t* inlier_idx; int32_t* pm_int;
    double* dp; double* Gpm; double* temp; int32_t* prev;
} emba_oracle_dump;

/* a1-a7: LEGM::evaluateDataError (src/emba/model.cpp:72-258) with eval_deriv=true.
 * events: x,y (sensor px), pol in {0,1}, t_ns sorted ascending.  ep_out capacity n.
 * num_ev_map: pano_h*pano_w int32, overwritten.  Returns the number of inliers, or -1 on error
 * (batch time outside the spline's knots). */
long emba_oracle_eval_data_error(emba_oracle* o, const double* knots_xyzw, int K, int64_t t0_ns,
                                 int64_t dt_ns, const double* Gx, const double* Gy,
                                 const uint16_t* x, const uint16_t* y, const uint8_t* pol,
                                 const int64_t* t_ns, size_t n, double* ep_out,
                                 int32_t* num_ev_map, const emba_oracle_dump* dump);

/* Index-level results only (pm, rounded pixels, num_ev_map, inlier count) with O(sensor pixels) working memory: used to
 * measure the pm_int / num_ev_map flip rate of the device path at 10 M - 100 M events.  pm_out, pm_int_out: 2n each or NULL. */
long emba_oracle_count_map(const emba_oracle* o, const double* knots_xyzw, int K, int64_t t0_ns, int64_t dt_ns,
                           const uint16_t* x, const uint16_t* y, const int64_t* t_ns, size_t n, double* pm_out,
                           int32_t* num_ev_map, int32_t* pm_int_out);

/* Threading mode of evaluateDataError / formNormalEq / count_map.  1 (default) = "ref": one thread, the reference's order —
 * the checker.  n > 1 = "omp": the same per-measurement arithmetic on n host threads, order-relaxed where sums meet
 * (SURVEY.md §8d asks for both as CPU baselines).  Only bench.py's cpu_baseline leg and tests switch it on. */
void emba_oracle_set_threads(int n);
int emba_oracle_get_threads(void);
int emba_oracle_max_threads(void);

/* a8-a10: LEGM::formNormalEq / formNormalEqIRLS (model.cpp:316-491 / 493-687), using the state
 * left by the last emba_oracle_eval_data_error.  irls: 0 quadratic, 1 huber, 2 cauchy.
 * A11 3K*3K col-major, b1 3K; active_idx capacity H*W (ascending pano index);
 * A22 P*4 (each block row-major [xx xy; xy yy]), b2 2P; A12 3K x 2P col-major or NULL.
 * Returns P (number of active pixels). */
long emba_oracle_form_normal_eq(emba_oracle* o, const double* ep, int K, const int32_t* num_ev_map,
                                int thres_valid_pixel, int irls, double a, double* A11, double* b1,
                                uint32_t* active_idx, double* A22, double* b2, double* A12);

/* a11: LEGM::applyL2Reg (model.cpp:689-719). */
void emba_oracle_apply_l2(const emba_oracle* o, size_t P, const uint32_t* active_idx, double alpha,
                          const double* Gx, const double* Gy, double* A22, double* b2);

/* f2: LEGM::updateMap (model.cpp:863-903), in place: active pixels += damping*x2, every other pixel := 0. */
void emba_oracle_update_map(size_t P, const uint32_t* active_idx, size_t npix, const double* x2, double damping,
                            double* Gx, double* Gy);

/* f1: LEGM::solveNormalEq (model.cpp:721-792), dense like the reference (small sizes only): A11 n x n and A12 n x 2P
 * column-major, A22 P x [xx xy; xy yy], LM damping A?m = A + lambda*diag(A).  x1[n], x2[2P].  The n x n solve is Eigen's pivoted
 * LDL^T restated (emba_oracle_ldlt_solve), the 2x2 inverses Eigen's Matrix2d::inverse (emba_oracle_inverse2).  Always produces x1 / x2
 * like the reference (a zero pivot gives a zero component, a singular 2x2 block inf / nan); returns what ldlt.info() would say
 * (0 Success, 1 NumericalIssue), which the reference never reads. */
int emba_oracle_solve_normal_eq(int n, size_t P, const double* A11, const double* A12, const double* A22, const double* b1,
                                const double* b2, double lambda, double* x1, double* x2);

/* The two Eigen calls of solveNormalEq, restated from the reference's vendored Eigen 3.3.9 and pinned against it (oracle/ref_eigen.cpp):
 * Matrix2d::inverse() (model.cpp:750; row-major 2x2 in and out) and S.ldlt().solve(rhs) (model.cpp:789; S n x n column-major, lower
 * triangle read and overwritten; vecD / transp may be NULL). */
void emba_oracle_inverse2(const double* A, double* out);
int emba_oracle_ldlt_solve(double* S, int n, const double* rhs, double* x, double* vecD, int* transp);

/* f1 on the sparse form of A12 (one rank-1 factor per measurement, taken from the state of the last evaluateDataError): the same
 * LEGM::solveNormalEq / solveNormalEqCG (model.cpp:721-792 / 794-840) for sizes where the dense 3K x 2P matrix does not fit.
 * A11 (3K)^2 col-major, b1 3K (untrimmed); A22 P x [xx xy; xy yy], b2 2P after applyL2Reg; skip = 3 holds the first control pose
 * fixed (solver.cpp:156-165).  x1[3K], x2[2P]. */
int emba_oracle_solve_sparse(emba_oracle* o, const double* ep, int K, const int32_t* num_ev_map, int thres, int irls, double a,
                             const double* A11, const double* b1, size_t P, const uint32_t* active, const double* A22,
                             const double* b2, double lambda, int skip, double* x1, double* x2);
int emba_oracle_solve_cg_sparse(emba_oracle* o, const double* ep, int K, const int32_t* num_ev_map, int thres, int irls, double a,
                                const double* A11, const double* b1, size_t P, const uint32_t* active, const double* A22,
                                const double* b2, double lambda, int skip, int max_iter, double tol, double* x1, double* x2,
                                int* iters_out, double* err_out);

/* a12: cost terms.  0.5*ep.ep (src/emba/solver.cpp:88); evaluateRobustDataCost (model.cpp:279-314);
 * 0.5*alpha*|evaluateRegError|^2 (model.cpp:260-277, solver.cpp:90). */
double emba_oracle_data_cost(const double* ep, size_t m, int irls, double a);
double emba_oracle_reg_cost(const double* Gx, const double* Gy, size_t npix, double alpha);

#ifdef __cplusplus
}
#endif
#endif
