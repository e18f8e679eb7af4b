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
// (round 6, measured and dropped: the four quotients by the same n with the reciprocal's refinement shared — the compiler's own fdiv expansion written out once,
// bit-identical in every parity test — saves ~35 of the tiled kernel's ~1100 VALU instructions per event: warp kernel -1.4 % at config 3, -1 % at 3 M, +1 us at 1 M;
// profiles/r06_shared_rcp_ab.txt.  Not worth a second path to the rounded pixel.)
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

// sin and cos of the half rotation angle.  Between neighbouring control poses it is far below pi/4, where the two polynomial kernels
// of fdlibm (k_sin.c / k_cos.c, |x| <= pi/4, < 1 ulp) need no argument reduction and no quadrant selection: a third of the instructions
// of the general sin() + cos(), which stay as the fallback.  (The oracle calls the host libm; neither device form is bit-identical to it,
// both are within an ulp — scripts/flip_rate.py measures what that does to round(pm).)
__device__ __forceinline__ void sin_cos_half(double x, double& sn, double& cs)
{
    if (fabs(x) <= 0x1.921fb54442d18p-1) {
        // (explicit fma: the same instructions wherever this is inlined, whatever the contraction mode)
        const double z = x * x;
        const double rs = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                     -1.98412698298579493134e-04), 8.33333333332248946124e-03);
        sn = fma(z * x, fma(z, rs, -1.66666666666666324348e-01), x);
        const double rc = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                              2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
        const double hz = 0.5 * z, w = 1.0 - hz;
        cs = w + fma(z, rc, (1.0 - w) - hz);
    } else {
        sn = sin(x); cs = cos(x);
    }
}

// Jacobian-only reciprocal: v_rcp_f64 and two Newton steps (about an ulp) instead of the IEEE division's ~14 instructions.  Never
// used for anything an integer result depends on; its arguments here are O(1) (no scaling for denormal or huge values needed).
__device__ __forceinline__ double rcp_nr(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
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
        double sn, cs;
        sin_cos_half(half, sn, cs);
        imag = sn / theta;
        real = cs;
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

// Tile order: the pose of an event is evaluated PER EVENT from per-SEGMENT constants, so the warp kernel gathers nothing per batch.
// (At 100 M events the batch table is 1 M records: every event pulled its own 128-B line from HBM — 13.6 GB per launch, 43 % of the
// kernel's traffic by the counters — while the VALU was a third busy.)  The K-1 segment records (96 B each) stay in cache.
//   seg[12] = { p0[4], delta[3], r1, k[3], r2 }   with delta = log(p0^-1 p1), theta = |delta|, k = R0 delta/theta (world-frame axis),
//   r1 = -theta/2, r2 = theta^2 c(theta): the coefficients of Jl^-1(delta) = I + r1 A + r2 A^2 in A = hat(delta/theta)
//   (sophus_utils.hpp:372-414, incl. its small-angle and near-pi branches).
// Value: q = p0 * exp(u delta), the SAME calls in the same order as So3Spline<2>::evaluate (so3_spline.h:233-247) — bit-identical to the
// per-batch evaluation, which is what round(pm) needs.  Jacobian: Jl(u delta) = I + p1 A + p2 A^2 (sophus_utils.hpp:332-362) commutes with
// Jl^-1(delta) and R0 A R0^T = hat(k), hence J1 = u R0 Jl Jl^-1 R0^T = u (I + alpha hat(k) + gamma hat(k)^2) (so3_spline.h:261-270) with
// alpha = p1 + r1 - p1 r2 - p2 r1, gamma = p2 + r2 + p1 r1 - p2 r2 (A^3 = -A): the reference's J1 to rounding (Jacobian tolerance only).
__device__ __forceinline__ void segment_consts(const double* p0, const double* p1, double* seg)
{
    double p0inv[4] = {-p0[0], -p0[1], -p0[2], p0[3]};
    quat_normalize(p0inv);
    double r01[4], delta[3], R0[9];
    so3_mul(p0inv, p1, r01);
    so3_log(r01, delta);
    const double n2 = sqn3(delta[0] * delta[0], delta[1] * delta[1], delta[2] * delta[2]);
    const double th = sqrt(n2);
    double k[3] = {0, 0, 0}, r2;
    if (th > 0.0) {
        quat_to_matrix(p0, R0);
        const double a0 = delta[0] / th, a1 = delta[1] / th, a2 = delta[2] / th;
#pragma unroll
        for (int r = 0; r < 3; ++r) k[r] = R0[3 * r] * a0 + R0[3 * r + 1] * a1 + R0[3 * r + 2] * a2;
    }
    if (n2 > kSophusEps) {
        if (th < kPi - sqrt(kSophusEps)) r2 = 1.0 - th * (1.0 + cos(th)) / (2.0 * sin(th));
        else r2 = n2 / (kPi * kPi);
    } else {
        r2 = n2 / 12.0;
    }
    seg[0] = p0[0]; seg[1] = p0[1]; seg[2] = p0[2]; seg[3] = p0[3];
    seg[4] = delta[0]; seg[5] = delta[1]; seg[6] = delta[2]; seg[7] = -0.5 * th;
    seg[8] = k[0]; seg[9] = k[1]; seg[10] = k[2]; seg[11] = r2;
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

// sin and cos of the half angle behind a REAL call (see project_angles_call below: fp64 polynomial constants hoisted out of the tiled kernel's loop)
__device__ __attribute__((noinline)) double2 sin_cos_call(double x)
{
    double sn, cs;
    sin_cos_half(x, sn, cs);
    return make_double2(sn, cs);
}

template <bool CALL>
__device__ __forceinline__ void spline2_event(const double* seg, double u, double* q_out, double* J1)
{
    // value: kdelta = delta * u; exp; p0 * exp  (the tail of spline2_eval, operation for operation)
    const double kd[3] = {seg[4] * u, seg[5] * u, seg[6] * u};
    const double theta_sq = sqn3(kd[0] * kd[0], kd[1] * kd[1], kd[2] * kd[2]);
    double imag, real, p1c, p2c;            // p1c, p2c: coefficients of Jl(u delta) (only the Jacobian uses them)
    if (theta_sq < kSophusEps * kSophusEps) {                          // so3.hpp:594
        const double theta_po4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
        real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
        const double n = sqrt(theta_sq);
        p1c = 0.5 * n; p2c = theta_sq * (1.0 / 6.0);
    } else {
        const double theta = sqrt(theta_sq);
        const double half = 0.5 * theta;
        double sh, ch;
        if (CALL) { const double2 sc = sin_cos_call(half); sh = sc.x; ch = sc.y; }
        else sin_cos_half(half, sh, ch);
        imag = sh / theta;
        real = ch;
        if (theta_sq > kSophusEps) { const double it2 = 2.0 * rcp_nr(theta); p1c = sh * sh * it2; p2c = 1.0 - sh * ch * it2; }     // (1 - cos n)/n, (n - sin n)/n
        else { p1c = 0.5 * theta; p2c = theta_sq * (1.0 / 6.0); }                                                    // sophus_utils.hpp:351 small branch
    }
    const double e[4] = {imag * kd[0], imag * kd[1], imag * kd[2], real};
    so3_mul(seg, e, q_out);
    const double r1 = seg[7], r2 = seg[11];
    const double alpha = p1c + r1 - p1c * r2 - p2c * r1, gamma = p2c + r2 + p1c * r1 - p2c * r2;
    rebuild_j1(u, u * alpha, u * gamma, seg + 8, J1);
}

__device__ __forceinline__ void project_jacobian(const double* rb, double fx, double fy, double* J23);

// EquirectangularCamera::projectToImage (include/utils/equirectangular_camera.h:18-45) chained with
// -[rb]x (event_pano_warper.cpp:62-65): pm (2) and J23 = dpm_drb * drb_ddrot (row-major 2x3).
// The projection alone, as a REAL function (noinline): its atan2 / asin carry ~40 fp64 polynomial constants, which the compiler
// hoists out of a loop and keeps in registers for the loop's whole life (fp64 literals cannot be encoded in VALU instructions
// on gfx9: the tiled warp kernel, whose waves loop over event groups, went from 88 to 190 VGPRs that way).  Behind a call the
// constants are materialised per call and die at the return.  Same expressions, same results as the inline form.
// (results by value: out-pointers to the caller's locals would travel through scratch memory)
__device__ __attribute__((noinline)) double2 project_angles_call(double x, double y, double z)
{
    const double phi = atan2(x, z);
    const double r2 = x * x + y * y + z * z;
    return make_double2(phi, asin(y / sqrt(r2)));
}


template <bool CALL = false>
__device__ __forceinline__ void project_chain(const double* rb, double fx, double fy, double cx, double cy,
                                              double* pm, double* J23)
{
    if (CALL) {
        const double2 a = project_angles_call(rb[0], rb[1], rb[2]);
        pm[0] = cx + a.x * fx;
        pm[1] = cy + a.y * fy;
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
    // (round 2: the reciprocals are rcp_nr, and sqrt(1 - (y/rho)^2) = sqrt(x^2 + z^2) / rho shares 1/(x^2 + z^2) with the first row)
    const double d2 = x * x + z * z;
    const double rho = sqrt(d2 + y * y);
    const double inv_rho = rcp_nr(rho);
    const double Ydivrho = y * inv_rho;
    const double inv_d2 = rcp_nr(d2);
    const double inv_d = fx * inv_d2;
    const double tmp2 = -fy * rho * (sqrt(d2) * inv_d2);
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
