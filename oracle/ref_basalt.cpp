// oracle/ref_basalt.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" driver around the REFERENCE's own, unmodified, header-only
// SO(3) spline code, compiled from where it lies under /root/reference:
//   thirdparty/basalt-headers/include/basalt/spline/so3_spline.h:218-274  (So3Spline<2>::evaluate)
//   thirdparty/basalt-headers/include/basalt/utils/sophus_utils.hpp:332-414 (leftJacobianSO3 / InvSO3)
//   thirdparty/basalt-headers/thirdparty/Sophus/sophus/so3.hpp           (exp/log/product/matrix)
//   thirdparty/basalt-headers/thirdparty/eigen                            (Eigen, vendored)
// Built by oracle/Makefile into oracle/_ref/libref_basalt.so (git-ignored). Used to pin
// row a3 of the oracle (oracle/emba_oracle.c: emba_oracle_spline_eval) and to generate
// tests/golden/so3_spline_n2.bin (tests/golden/make_so3_spline_golden.py).
//
// The wrapper mirrors how the reference calls the spline:
//   LinearTrajectory ctor            src/utils/trajectory.cpp:59-73  (dt_ns, t0_ns, knotsPushBack)
//   LinearTrajectory::evaluate       src/utils/trajectory.cpp:122-147 (3x6 packing [J0 | J1])
#include <basalt/spline/so3_spline.h>
#include <cstdint>

extern "C" {

// knots_xyzw: K unit quaternions (x,y,z,w). Returns 0 on success.
// out_q_xyzw[4], out_R[9] row-major (Sophus::SO3d::matrix()), out_J36[18] row-major 3x6.
int ref_so3spline2_evaluate(const double* knots_xyzw, int K, int64_t t0_ns, int64_t dt_ns,
                            int64_t t_ns, double* out_q_xyzw, double* out_R, int* out_start_idx,
                            double* out_J36)
{
    basalt::So3Spline<2, double> spline(dt_ns, t0_ns);
    for (int i = 0; i < K; ++i) {
        Eigen::Quaterniond q(knots_xyzw[4 * i + 3], knots_xyzw[4 * i + 0], knots_xyzw[4 * i + 1],
                             knots_xyzw[4 * i + 2]);
        spline.knotsPushBack(Sophus::SO3d(q));
    }
    const int64_t st = t_ns - t0_ns;
    if (st < 0 || st / dt_ns + 2 > K) return 1;
    basalt::So3Spline<2, double>::JacobianStruct J;
    Sophus::SO3d R = spline.evaluate(t_ns, &J);
    const Eigen::Quaterniond& q = R.unit_quaternion();
    out_q_xyzw[0] = q.x(); out_q_xyzw[1] = q.y(); out_q_xyzw[2] = q.z(); out_q_xyzw[3] = q.w();
    Eigen::Matrix3d M = R.matrix();
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) out_R[3 * r + c] = M(r, c);
    *out_start_idx = (int)J.start_idx;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            out_J36[6 * r + c]     = J.d_val_d_knot[0](r, c);
            out_J36[6 * r + c + 3] = J.d_val_d_knot[1](r, c);
        }
    return 0;
}

// Sophus::SO3d::exp / log and basalt's left Jacobians, for branch-level pinning.
void ref_so3_exp(const double* w, double* out_q_xyzw)
{
    Sophus::SO3d R = Sophus::SO3d::exp(Eigen::Vector3d(w[0], w[1], w[2]));
    const Eigen::Quaterniond& q = R.unit_quaternion();
    out_q_xyzw[0] = q.x(); out_q_xyzw[1] = q.y(); out_q_xyzw[2] = q.z(); out_q_xyzw[3] = q.w();
}
void ref_so3_log(const double* q_xyzw, double* out_w)
{
    Eigen::Quaterniond q(q_xyzw[3], q_xyzw[0], q_xyzw[1], q_xyzw[2]);
    Eigen::Vector3d w = Sophus::SO3d(q).log();
    out_w[0] = w[0]; out_w[1] = w[1]; out_w[2] = w[2];
}
void ref_left_jacobian(const double* phi, double* out_J, double* out_Jinv)
{
    Eigen::Matrix3d J, Ji;
    Eigen::Vector3d p(phi[0], phi[1], phi[2]);
    Sophus::leftJacobianSO3(p, J);
    Sophus::leftJacobianInvSO3(p, Ji);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { out_J[3 * r + c] = J(r, c); out_Jinv[3 * r + c] = Ji(r, c); }
}
// Group product / inverse / matrix, for step-level pinning of the quaternion path.
static Sophus::SO3d from_xyzw(const double* q)
{
    Sophus::SO3d R;  // identity
    R = Sophus::SO3d(Eigen::Quaterniond(q[3], q[0], q[1], q[2]));
    return R;
}
void ref_so3_mul(const double* a_xyzw, const double* b_xyzw, double* out_xyzw)
{
    Sophus::SO3d r = from_xyzw(a_xyzw) * from_xyzw(b_xyzw);
    const Eigen::Quaterniond& q = r.unit_quaternion();
    out_xyzw[0] = q.x(); out_xyzw[1] = q.y(); out_xyzw[2] = q.z(); out_xyzw[3] = q.w();
}
void ref_so3_inverse(const double* a_xyzw, double* out_xyzw)
{
    Sophus::SO3d r = from_xyzw(a_xyzw).inverse();
    const Eigen::Quaterniond& q = r.unit_quaternion();
    out_xyzw[0] = q.x(); out_xyzw[1] = q.y(); out_xyzw[2] = q.z(); out_xyzw[3] = q.w();
}
void ref_so3_matrix(const double* a_xyzw, double* out_R)
{
    Eigen::Matrix3d M = from_xyzw(a_xyzw).matrix();
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) out_R[3 * r + c] = M(r, c);
}
}
