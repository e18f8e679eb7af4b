// emba_amd/csrc/device_math.h — gfx950 device functions for the geometry leaves of the EMBA hot path.
// All IEEE double, written from the formulas of the reference (file:line cited per function); the
// evaluation order follows the reference where that is cheap, but only the integer results
// (control-pose index, rounded panorama pixel) are required to be bit-exact (BASELINE north_star).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace emba {

// Everything in this header up to the "Jacobian only" part of project_chain feeds an INTEGER the reference computes — the
// control-pose index and the rounded panorama pixel round(pm) (model.cpp:209-211) — so it is compiled WITHOUT floating-point
// contraction: the oracle is built -ffp-contract=off (oracle/Makefile) and a fused multiply-add here would make the two
// disagree in the last bit of pm, i.e. leave bit-exact indexing to luck.  With contraction off every operation below is the
// same IEEE operation in the same order as in the oracle; what remains is libm (ocml vs glibc atan2/asin/sin/cos/atan).
#pragma clang fp contract(off)

constexpr double kSophusEps = 1e-10;  // Sophus::Constants<double>::epsilon(), sophus/common.hpp:94
constexpr double kPi = 3.14159265358979323846;

__device__ __forceinline__ double sum3(double a, double b, double c) { return a + (b + c); }
__device__ __forceinline__ double sqn3(double a, double b, double c) { return (a + b) + c; }

// C = A*B, row-major 3x3 (same summation pattern as Eigen's 3x3 lazy product into a Matrix3d:
// rows 0-1 ((t0+t1)+t2), row 2 (t0+(t1+t2)) — see oracle/emba_oracle.c mat3_mul_seq).
__device__ __forceinline__ void mat3_mul(const double* A, const double* B, double* C)
{
    double T[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double t0 = A[3 * i + 0] * B[0 + j], t1 = A[3 * i + 1] * B[3 + j], t2 = A[3 * i + 2] * B[6 + j];
            T[3 * i + j] = (i < 2) ? (t0 + t1) + t2 : t0 + (t1 + t2);
        }
#pragma unroll
    for (int i = 0; i < 9; ++i) C[i] = T[i];
}

__device__ __forceinline__ void hat3(const double* p, double* M)
{   // Sophus::SO3::hat
    M[0] = 0;     M[1] = -p[2]; M[2] = p[1];
    M[3] = p[2];  M[4] = 0;     M[5] = -p[0];
    M[6] = -p[1]; M[7] = p[0];  M[8] = 0;
}

// SO3Base::normalize (sophus/so3.hpp:297-303); (x,y,z,w) storage.
__device__ __forceinline__ void quat_normalize(double* q)
{
    const double n2 = (q[0] * q[0] + q[2] * q[2]) + (q[1] * q[1] + q[3] * q[3]);
    const double n = sqrt(n2);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

// SO3Base::operator* (so3.hpp:324-339) followed by the normalizing SO3(Quaternion) ctor (:481-487).
__device__ __forceinline__ void so3_mul(const double* a, const double* b, double* out)
{
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    double r[4];
    r[3] = aw * bw - ax * bx - ay * by - az * bz;
    r[0] = aw * bx + ax * bw + ay * bz - az * by;
    r[1] = aw * by + ay * bw + az * bx - ax * bz;
    r[2] = aw * bz + az * bw + ax * by - ay * bx;
    quat_normalize(r);
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2]; out[3] = r[3];
}

// Eigen QuaternionBase::toRotationMatrix via SO3Base::matrix() (so3.hpp:310-312); row-major.
__device__ __forceinline__ void quat_to_matrix(const double* q, double* R)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// Sophus::SO3::expAndTheta (so3.hpp:583-619)
__device__ __forceinline__ void so3_exp(const double* w, double* q)
{
    const double theta_sq = sqn3(w[0] * w[0], w[1] * w[1], w[2] * w[2]);
    double imag, real;
    if (theta_sq < kSophusEps * kSophusEps) {
        const double theta_po4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
        real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
    } else {
        const double theta = sqrt(theta_sq);
        const double half = 0.5 * theta;
        imag = sin(half) / theta;
        real = cos(half);
    }
    q[0] = imag * w[0]; q[1] = imag * w[1]; q[2] = imag * w[2]; q[3] = real;
}

// Sophus::SO3Base::logAndTheta (so3.hpp:247-290)
__device__ __forceinline__ void so3_log(const double* q, double* w)
{
    const double squared_n = sqn3(q[0] * q[0], q[1] * q[1], q[2] * q[2]);
    const double qw = q[3];
    double f;
    if (squared_n < kSophusEps * kSophusEps) {
        const double squared_w = qw * qw;
        f = 2.0 / qw - (2.0 / 3.0) * (squared_n) / (qw * squared_w);
    } else {
        const double n = sqrt(squared_n);
        if (fabs(qw) < kSophusEps) f = (qw > 0.0) ? kPi / n : -kPi / n;
        else f = 2.0 * atan(n / qw) / n;
    }
    w[0] = f * q[0]; w[1] = f * q[1]; w[2] = f * q[2];
}

// Sophus::leftJacobianSO3 (basalt/utils/sophus_utils.hpp:332-362)
__device__ __forceinline__ void left_jacobian(const double* phi, double* J)
{
    const double n2 = sqn3(phi[0] * phi[0], phi[1] * phi[1], phi[2] * phi[2]);
    double H[9], H2[9];
    hat3(phi, H);
    mat3_mul(H, H, H2);
#pragma unroll
    for (int i = 0; i < 9; ++i) J[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (n2 > kSophusEps) {
        const double n = sqrt(n2);
        const double n3 = n2 * n;
        const double c1 = 1 - cos(n), c2 = n - sin(n);
#pragma unroll
        for (int i = 0; i < 9; ++i) J[i] += H[i] * c1 / n2;
#pragma unroll
        for (int i = 0; i < 9; ++i) J[i] += H2[i] * c2 / n3;
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) J[i] += H[i] / 2;
#pragma unroll
        for (int i = 0; i < 9; ++i) J[i] += H2[i] / 6;
    }
}

// Sophus::leftJacobianInvSO3 (basalt/utils/sophus_utils.hpp:372-414)
__device__ __forceinline__ void left_jacobian_inv(const double* phi, double* J)
{
    const double n2 = sqn3(phi[0] * phi[0], phi[1] * phi[1], phi[2] * phi[2]);
    double H[9], H2[9];
    hat3(phi, H);
    mat3_mul(H, H, H2);
#pragma unroll
    for (int i = 0; i < 9; ++i) J[i] = ((i % 4 == 0) ? 1.0 : 0.0) - H[i] / 2;
    if (n2 > kSophusEps) {
        const double n = sqrt(n2);
        if (n < kPi - sqrt(kSophusEps)) {
            const double c = 1 / n2 - (1 + cos(n)) / (2 * n * sin(n));
#pragma unroll
            for (int i = 0; i < 9; ++i) J[i] += H2[i] * c;
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) J[i] += H2[i] / (kPi * kPi);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) J[i] += H2[i] / 12;
    }
}

// basalt::So3Spline<2>::evaluate (so3_spline.h:218-274) for one query; knots (x,y,z,w).
// Outputs the value as unit quaternion q (x,y,z,w) and J1 = d_val_d_knot[1] (row-major); d_val_d_knot[0] = I - J1.
__device__ __forceinline__ void spline2_eval(const double* p0, const double* p1, double u, double* q_out, double* J1)
{
    double p0inv[4] = {-p0[0], -p0[1], -p0[2], p0[3]};
    quat_normalize(p0inv);                       // SO3::inverse -> SO3(conjugate) normalizes
    double r01[4], delta[3], kdelta[3];
    so3_mul(p0inv, p1, r01);
    so3_log(r01, delta);
#pragma unroll
    for (int i = 0; i < 3; ++i) kdelta[i] = delta[i] * u;
    double Jli[9], Jlk[9], T[9], R0[9], R0inv[9];
    left_jacobian_inv(delta, Jli);
    left_jacobian(kdelta, Jlk);
    quat_to_matrix(p0, R0);
#pragma unroll
    for (int i = 0; i < 9; ++i) T[i] = u * R0[i];
    mat3_mul(T, Jlk, T);
    mat3_mul(T, Jli, T);
    quat_to_matrix(p0inv, R0inv);
    mat3_mul(T, R0inv, J1);
    double e[4], res[4];
    so3_exp(kdelta, e);
    so3_mul(p0, e, res);
    q_out[0] = res[0]; q_out[1] = res[1]; q_out[2] = res[2]; q_out[3] = res[3];
}

__device__ __forceinline__ void project_jacobian(const double* rb, double fx, double fy, double* J23);

// Compact form of d_val_d_knot[1] (tile order: the per-batch pose record shrinks from 112 B to one 64-B line).
// J1 = u R0 Jl(u delta) Jl^-1(delta) R0^T (so3_spline.h:261-270).  Jl(u delta) and Jl^-1(delta) are polynomials in hat(delta)
// (sophus_utils.hpp:332-414) and R0 hat(a) R0^T = hat(R0 a), so J1 = u I + a1 hat(k) + a2 hat(k)^2 with k = R0 delta/|delta| a
// property of the spline SEGMENT and (u, a1, a2) of the batch.  a1, a2 are taken by projecting the reference-order J1 of the batch
// onto hat(k) and hat(k)^2, so the per-event reconstruction returns that J1 to rounding (1e-16 relative: Jacobian tolerance only,
// nothing index-level depends on J1).  k = 0 for coinciding knots (then J1 = u I).
__device__ __forceinline__ void segment_axis(const double* p0, const double* p1, double* k)
{
    double p0inv[4] = {-p0[0], -p0[1], -p0[2], p0[3]};
    quat_normalize(p0inv);
    double r01[4], delta[3], R0[9];
    so3_mul(p0inv, p1, r01);
    so3_log(r01, delta);
    const double th = sqrt(sqn3(delta[0] * delta[0], delta[1] * delta[1], delta[2] * delta[2]));
    if (!(th > 0.0)) { k[0] = 0; k[1] = 0; k[2] = 0; return; }
    quat_to_matrix(p0, R0);
    const double a0 = delta[0] / th, a1 = delta[1] / th, a2 = delta[2] / th;
#pragma unroll
    for (int r = 0; r < 3; ++r) k[r] = R0[3 * r] * a0 + R0[3 * r + 1] * a1 + R0[3 * r + 2] * a2;
}

// (u, a1, a2) of a batch from its exact J1 and the segment axis k
__device__ __forceinline__ void project_j1(const double* J1, double u, const double* k, double& a1, double& a2)
{
    const double kk = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
    a1 = 0; a2 = 0;
    if (!(kk > 0.0)) return;
    // antisymmetric part = hat(w), w = (J21 - J12, J02 - J20, J10 - J01) / 2  =>  a1 = w.k / k.k
    const double w0 = 0.5 * (J1[7] - J1[5]), w1 = 0.5 * (J1[2] - J1[6]), w2 = 0.5 * (J1[3] - J1[1]);
    a1 = (w0 * k[0] + w1 * k[1] + w2 * k[2]) / kk;
    // symmetric part - u I = a2 N, N = k k^T - (k.k) I
    double num = 0, den = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double n = k[r] * k[c] - ((r == c) ? kk : 0.0);
            const double m = 0.5 * (J1[3 * r + c] + J1[3 * c + r]) - ((r == c) ? u : 0.0);
            num += m * n; den += n * n;
        }
    a2 = (den > 0.0) ? num / den : 0.0;
}

// J1 = u I + a1 hat(k) + a2 (k k^T - (k.k) I)
__device__ __forceinline__ void rebuild_j1(double u, double a1, double a2, const double* k, double* J1)
{
    const double kk = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
    const double d = u - a2 * kk;
    J1[0] = d + a2 * k[0] * k[0]; J1[1] = a2 * k[0] * k[1] - a1 * k[2]; J1[2] = a2 * k[0] * k[2] + a1 * k[1];
    J1[3] = a2 * k[1] * k[0] + a1 * k[2]; J1[4] = d + a2 * k[1] * k[1]; J1[5] = a2 * k[1] * k[2] - a1 * k[0];
    J1[6] = a2 * k[2] * k[0] - a1 * k[1]; J1[7] = a2 * k[2] * k[1] + a1 * k[0]; J1[8] = d + a2 * k[2] * k[2];
}

// EquirectangularCamera::projectToImage (include/utils/equirectangular_camera.h:18-45) chained with
// -[rb]x (event_pano_warper.cpp:62-65): pm (2) and J23 = dpm_drb * drb_ddrot (row-major 2x3).
// The projection alone, as a REAL function (noinline): its atan2 / asin carry ~40 fp64 polynomial constants, which the compiler
// hoists out of a loop and keeps in registers for the loop's whole life (fp64 literals cannot be encoded in VALU instructions
// on gfx9: the tiled warp kernel, whose waves loop over event groups, went from 88 to 190 VGPRs that way).  Behind a call the
// constants are materialised per call and die at the return.  Same expressions, same results as the inline form.
__device__ __attribute__((noinline)) void project_angles_call(double x, double y, double z, double* phi, double* theta)
{
    *phi = atan2(x, z);
    const double r2 = x * x + y * y + z * z;
    *theta = asin(y / sqrt(r2));
}

template <bool CALL = false>
__device__ __forceinline__ void project_chain(const double* rb, double fx, double fy, double cx, double cy,
                                              double* pm, double* J23)
{
    if (CALL) {
        double phi, theta;
        project_angles_call(rb[0], rb[1], rb[2], &phi, &theta);
        pm[0] = cx + phi * fx;
        pm[1] = cy + theta * fy;
        project_jacobian(rb, fx, fy, J23);
        return;
    }
    const double x = rb[0], y = rb[1], z = rb[2];
    const double phi = atan2(x, z);
    const double r2 = x * x + y * y + z * z;
    const double theta = asin(y / sqrt(r2));
    pm[0] = cx + phi * fx;
    pm[1] = cy + theta * fy;
    project_jacobian(rb, fx, fy, J23);
}

// Jacobian only from here on: contraction is allowed again (last-bit differences, tolerance 1e-5 relative in the north star).
#pragma clang fp contract(fast)
__device__ __forceinline__ void project_jacobian(const double* rb, double fx, double fy, double* J23)
{
    const double x = rb[0], y = rb[1], z = rb[2];
    // The Jacobian row formulas of equirectangular_camera.h:28-41 with their seven divisions folded into three reciprocals
    // (fp64 division is ~25 VALU instructions here and this kernel's arithmetic is not free):
    //   tmp1 = fx / ((1 + (x/z)^2) z) = fx z / (x^2 + z^2),   tmp1 * (x/z) = fx x / (x^2 + z^2),
    //   tmp2 = -fy / sqrt(1 - (y/rho)^2),  tmp3 = (y/rho) / rho^2  with 1/rho taken once.
    // Only the Jacobian is affected (last-bit differences); phi and theta — the rounded pixel — use the reference's expressions.
    const double rho = sqrt(sum3(x * x, y * y, z * z));
    const double inv_rho = 1.0 / rho;
    const double Ydivrho = y * inv_rho;
    const double inv_d = fx / (x * x + z * z);
    const double tmp2 = -fy / sqrt(1 - Ydivrho * Ydivrho);
    const double tmp3 = Ydivrho * inv_rho * inv_rho;
    const double Jp[6] = {z * inv_d, 0.0, -x * inv_d, tmp2 * tmp3 * x, tmp2 * (tmp3 * y - inv_rho), tmp2 * tmp3 * z};
    const double M[9] = {0, rb[2], -rb[1], -rb[2], 0, rb[0], rb[1], -rb[0], 0};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += Jp[3 * i + k] * M[3 * k + j];
            J23[3 * i + j] = s;
        }
}

}  // namespace emba
