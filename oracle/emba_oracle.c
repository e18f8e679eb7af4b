/* oracle/emba_oracle.c — TEST INFRASTRUCTURE. NOT PART OF THE PRODUCT.
 * See oracle/emba_oracle.h for scope and pinning status ("a3 pinned; rest PARITY UNPINNED").
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 */
#define _GNU_SOURCE
#include "emba_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Two modes: "ref" = ONE thread, the reference's own evaluation order (the reference is single-threaded: no
 * `#pragma omp` anywhere despite -fopenmp in CMakeLists.txt:7) — this is the checker; "omp" = the same per-measurement
 * arithmetic spread over the host's cores, order-relaxed where sums meet (A11/b1 per-thread then merged in thread order,
 * A22/b2 by atomic adds) — only bench.py's cpu_baseline leg and the test that compares the two modes switch it on. */
static int g_threads = 1;
void emba_oracle_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int emba_oracle_get_threads(void) { return g_threads; }
int emba_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

#define SOPHUS_EPS 1e-10 /* Sophus::Constants<double>::epsilon(), sophus/common.hpp:94 */
#define EVENT_BATCH 100  /* model.cpp:78 (hard-coded, quirk Q2) */
#define OUTLIER_PX 10.0  /* model.cpp:200 */

/* ----------------------------------------------------------------------------------------
 * small fixed-size linear algebra, in Eigen's evaluation order for 3-term sums: x0 + (x1 + x2)
 * (Eigen/src/Core/Redux.h redux_novec_unroller splits [0,3) into [0,1) and [1,3)).
 * -------------------------------------------------------------------------------------- */
static inline double sum3(double a, double b, double c) { return a + (b + c); }
static inline double sqn3(double a, double b, double c) { return (a + b) + c; }

static void mat3_mul(const double* A, const double* B, double* C)
{
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            T[3 * i + j] = sum3(A[3 * i + 0] * B[0 + j], A[3 * i + 1] * B[3 + j], A[3 * i + 2] * B[6 + j]);
    memcpy(C, T, sizeof T);
}

/* 3x3 * 3x3 lazy products assigned to a column-major Matrix3d (Eigen 3.3.9, SSE2, EIGEN_UNALIGNED_VECTORIZE):
 * rows 0-1 of every column go through the packet path (etor_product_packet_impl: ((a0b0 + a1b1) + a2b2)),
 * row 2 through coeff() = redux_novec_unroller (a0b0 + (a1b1 + a2b2)).  Verified bit-for-bit against
 * oracle/_ref (tests/test_oracle_pinned.py). */
static void mat3_mul_seq(const double* A, const double* B, double* C)
{
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double t0 = A[3 * i + 0] * B[0 + j], t1 = A[3 * i + 1] * B[3 + j], t2 = A[3 * i + 2] * B[6 + j];
            T[3 * i + j] = (i < 2) ? (t0 + t1) + t2 : t0 + (t1 + t2);
        }
    memcpy(C, T, sizeof T);
}

static void hat3(const double* p, double* M)
{ /* Sophus::SO3::hat, so3.hpp:640-650 */
    M[0] = 0;     M[1] = -p[2]; M[2] = p[1];
    M[3] = p[2];  M[4] = 0;     M[5] = -p[0];
    M[6] = -p[1]; M[7] = p[0];  M[8] = 0;
}

/* ---- unit quaternion helpers, storage (x,y,z,w) like Eigen::Quaterniond::coeffs() ---- */

/* SO3Base::normalize, so3.hpp:297-303: coeffs() /= coeffs().norm().
 * Eigen evaluates the 4-element squaredNorm with SSE2 packets: (x^2+z^2) + (y^2+w^2). */
static void quat_normalize(double* q)
{
    double n2 = (q[0] * q[0] + q[2] * q[2]) + (q[1] * q[1] + q[3] * q[3]);
    double n = sqrt(n2);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

/* SO3Base::operator*, so3.hpp:324-339 (result goes through SO3(Quaternion) which normalizes, :481-487) */
static void so3_mul(const double* a, const double* b, double* out)
{
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    double r[4];
    r[3] = aw * bw - ax * bx - ay * by - az * bz;
    r[0] = aw * bx + ax * bw + ay * bz - az * by;
    r[1] = aw * by + ay * bw + az * bx - ax * bz;
    r[2] = aw * bz + az * bw + ax * by - ay * bx;
    quat_normalize(r);
    memcpy(out, r, sizeof r);
}

/* SO3Base::inverse, so3.hpp:229-231: SO3(conjugate) — the constructor normalizes again. */
static void so3_inverse(const double* q, double* out)
{
    double r[4] = {-q[0], -q[1], -q[2], q[3]};
    quat_normalize(r);
    memcpy(out, r, sizeof r);
}

/* Eigen QuaternionBase::toRotationMatrix (Eigen/src/Geometry/Quaternion.h:554-586),
 * reached through SO3Base::matrix(), so3.hpp:310-312.  Row-major output. */
static void quat_to_matrix(const double* q, double* R)
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

/* Sophus::SO3::expAndTheta, so3.hpp:583-619 */
void emba_oracle_so3_exp(const double* w, double* q)
{
    const double theta_sq = sqn3(w[0] * w[0], w[1] * w[1], w[2] * w[2]);
    double imag, real;
    if (theta_sq < SOPHUS_EPS * SOPHUS_EPS) {
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

/* Sophus::SO3Base::logAndTheta, so3.hpp:247-290 (input quaternion taken as stored) */
static void so3_log_raw(const double* q, double* w)
{
    const double squared_n = sqn3(q[0] * q[0], q[1] * q[1], q[2] * q[2]);
    const double qw = q[3];
    double two_atan_nbyw_by_n;
    if (squared_n < SOPHUS_EPS * SOPHUS_EPS) {
        const double squared_w = qw * qw;
        two_atan_nbyw_by_n = 2.0 / qw - (2.0 / 3.0) * (squared_n) / (qw * squared_w);
    } else {
        const double n = sqrt(squared_n);
        if (fabs(qw) < SOPHUS_EPS) {
            two_atan_nbyw_by_n = (qw > 0.0) ? M_PI / n : -M_PI / n;
        } else {
            two_atan_nbyw_by_n = 2.0 * atan(n / qw) / n;
        }
    }
    w[0] = two_atan_nbyw_by_n * q[0];
    w[1] = two_atan_nbyw_by_n * q[1];
    w[2] = two_atan_nbyw_by_n * q[2];
}

void emba_oracle_so3_log(const double* q_xyzw, double* w)
{
    double q[4] = {q_xyzw[0], q_xyzw[1], q_xyzw[2], q_xyzw[3]};
    quat_normalize(q); /* SO3(Quaternion) constructor, so3.hpp:481-487 */
    so3_log_raw(q, w);
}

/* Sophus::leftJacobianSO3 / leftJacobianInvSO3, basalt/utils/sophus_utils.hpp:332-362, 372-414 */
void emba_oracle_left_jacobian(const double* phi, double* J, double* Ji)
{
    const double n2 = sqn3(phi[0] * phi[0], phi[1] * phi[1], phi[2] * phi[2]);
    double H[9], H2[9];
    hat3(phi, H);
    mat3_mul(H, H, H2);
    static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};

    if (J) {
        memcpy(J, I3, sizeof I3);
        if (n2 > SOPHUS_EPS) {
            const double n = sqrt(n2);
            const double n3 = n2 * n;
            const double c1 = (1 - cos(n));
            const double c2 = (n - sin(n));
            /* Eigen: J += phi_hat * (1-cos) / n2 evaluates (phi_hat*s1)/s2 coefficient-wise */
            for (int i = 0; i < 9; ++i) J[i] += H[i] * c1 / n2;
            for (int i = 0; i < 9; ++i) J[i] += H2[i] * c2 / n3;
        } else {
            for (int i = 0; i < 9; ++i) J[i] += H[i] / 2;
            for (int i = 0; i < 9; ++i) J[i] += H2[i] / 6;
        }
    }
    if (Ji) {
        memcpy(Ji, I3, sizeof I3);
        for (int i = 0; i < 9; ++i) Ji[i] -= H[i] / 2;
        if (n2 > SOPHUS_EPS) {
            const double n = sqrt(n2);
            if (n < M_PI - sqrt(SOPHUS_EPS)) {
                const double c = (1 / n2 - (1 + cos(n)) / (2 * n * sin(n)));
                for (int i = 0; i < 9; ++i) Ji[i] += H2[i] * c;
            } else {
                for (int i = 0; i < 9; ++i) Ji[i] += H2[i] / (M_PI * M_PI);
            }
        } else {
            for (int i = 0; i < 9; ++i) Ji[i] += H2[i] / 12;
        }
    }
}

/* a2 — ros::Time/Duration midpoint (src/emba/model.cpp:116-119); rostime semantics per
 * Time-Time -> Duration::fromNSec; Duration*double -> Duration(toSec()*s);
 * Duration(double) = fromSec: sec=floor(d), nsec=round((d-sec)*1e9) half away from zero, carry;
 * Time+Duration adds fields and normalizes.  [parity unpinned: rostime is not in the reference tree] */
int64_t emba_oracle_batch_mid_ns(int64_t t_first_ns, int64_t t_last_ns)
{
    const int64_t d_ns = t_last_ns - t_first_ns;
    int64_t dsec = d_ns / 1000000000LL;
    int64_t dnsec = d_ns % 1000000000LL;
    if (dnsec < 0) { dnsec += 1000000000LL; dsec -= 1; } /* normalizeSecNSecSigned */
    const double dsecs = (double)dsec + 1e-9 * (double)dnsec; /* Duration::toSec */
    const double half = dsecs * 0.5;
    int64_t hsec = (int64_t)floor(half);
    int64_t hnsec = (int64_t)round((half - (double)hsec) * 1e9);
    hsec += hnsec / 1000000000LL;
    hnsec = hnsec % 1000000000LL;
    return t_first_ns + hsec * 1000000000LL + hnsec;
}

/* a3 — LinearTrajectory::evaluate (src/utils/trajectory.cpp:122-147) ->
 * basalt::So3Spline<2>::evaluate (so3_spline.h:218-274), DEG = 1, coeff = [1, u]
 * (spline_common.h:69-100 with N=2, cumulative). */
int emba_oracle_spline_eval(const double* knots, int K, int64_t t0_ns, int64_t dt_ns, int64_t t_ns,
                            double* q_out, double* R_out, int* cp_idx, double* J36)
{
    const int64_t st_ns = t_ns - t0_ns;
    if (st_ns < 0) return 1;
    const int64_t s = st_ns / dt_ns;
    const double u = (double)(st_ns % dt_ns) / (double)dt_ns;
    if ((size_t)(s + 2) > (size_t)K) return 1;

    const double* p0 = knots + 4 * s;
    const double* p1 = knots + 4 * (s + 1);
    double p0inv[4], r01[4], delta[3], kdelta[3];
    so3_inverse(p0, p0inv);
    so3_mul(p0inv, p1, r01);
    so3_log_raw(r01, delta);
    for (int i = 0; i < 3; ++i) kdelta[i] = delta[i] * u;

    double Jl_inv_delta[9], Jl_k_delta[9];
    emba_oracle_left_jacobian(delta, NULL, Jl_inv_delta);
    emba_oracle_left_jacobian(kdelta, Jl_k_delta, NULL);

    /* J_helper = coeff * res.matrix() * Jl_k_delta * Jl_inv_delta * p0.inverse().matrix() */
    double Rres[9], T[9], R0inv[9], Jh[9];
    quat_to_matrix(p0, Rres);
    for (int i = 0; i < 9; ++i) T[i] = u * Rres[i];
    mat3_mul_seq(T, Jl_k_delta, T);
    mat3_mul_seq(T, Jl_inv_delta, T);
    quat_to_matrix(p0inv, R0inv);
    mat3_mul_seq(T, R0inv, Jh);

    /* res *= exp(kdelta) */
    double e[4], res[4];
    emba_oracle_so3_exp(kdelta, e);
    so3_mul(p0, e, res);

    if (q_out) memcpy(q_out, res, sizeof res);
    if (R_out) quat_to_matrix(res, R_out);
    if (cp_idx) *cp_idx = (int)s;
    if (J36) {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                const double id = (r == c) ? 1.0 : 0.0;
                J36[6 * r + c] = id - Jh[3 * r + c];   /* d_val_d_knot[0] = I - J_helper */
                J36[6 * r + c + 3] = Jh[3 * r + c];    /* d_val_d_knot[1] = J_helper     */
            }
    }
    return 0;
}

/* a5 — EquirectangularCamera::projectToImage (include/utils/equirectangular_camera.h:18-45),
 * focalFromFOV(:64-67) with hfov=360, vfov=180 (event_pano_warper.cpp:19). */
void emba_oracle_project(int pano_w, int pano_h, const double* P, double* pm, double* J)
{
    const double fx = (double)((pano_w / 360.0) * 180.0 / M_PI);
    const double fy = (double)((pano_h / 180.0) * 180.0 / M_PI);
    const double cx = (double)pano_w / 2.0, cy = (double)pano_h / 2.0;
    const double x = P[0], y = P[1], z = P[2];
    const double phi = atan2(x, z);
    const double theta = asin(y / sqrt(x * x + y * y + z * z));
    const double rho = sqrt(sum3(x * x, y * y, z * z)); /* Eigen P.norm() */
    const double Ydivrho = y / rho;
    if (J) {
        const double XdivZ = x / z;
        const double tmp1 = fx / ((1 + XdivZ * XdivZ) * z);
        const double tmp2 = -fy / sqrt(1 - Ydivrho * Ydivrho);
        const double tmp3 = Ydivrho / (rho * rho);
        J[0] = tmp1;            J[1] = 0;                          J[2] = -tmp1 * XdivZ;
        J[3] = tmp2 * tmp3 * x; J[4] = tmp2 * (tmp3 * y - 1 / rho); J[5] = tmp2 * tmp3 * z;
    }
    pm[0] = cx + phi * fx;
    pm[1] = cy + theta * fy;
}

struct emba_oracle {
    int sw, sh, W, H;
    double C_th;
    double* lut; /* S*3 */
    /* state left by evaluateDataError (EventMap<State_LEGM>, event_map.h:22-113), kept as flat
     * arrays in ORIGINAL event order plus the pixel-major visiting order. */
    size_t n_used;      /* events actually warped (floor(n/100)*100, quirk Q1) */
    size_t cap;
    uint32_t* order;    /* pixel-major (y outer, x inner), then insertion (time) order */
    uint32_t* pix_start;/* S+1 offsets into order */
    uint32_t* pix;      /* sensor pixel index per event */
    double* pm;         /* 2 per event */
    double* D;          /* 12 per event: dpm_ddrot_cp row-major 2x6 */
    int32_t* cp;        /* per event */
    int32_t* inl;       /* inlier_idx per event */
    uint8_t* polv;
    double* dp; double* Gpm; double* temp; /* 2 per event */
    size_t first_counted; /* events before this index are warped and can be predecessors but form no measurement of their own
                           * (how a time shard sees the events in front of it, SURVEY.md §8e "halo"); 0 = the reference */
};

emba_oracle* emba_oracle_create(int sensor_w, int sensor_h, int pano_w, int pano_h,
                                const double* lut, double C_th)
{
    emba_oracle* o = (emba_oracle*)calloc(1, sizeof *o);
    if (!o) return NULL;
    o->sw = sensor_w; o->sh = sensor_h; o->W = pano_w; o->H = pano_h; o->C_th = C_th;
    const size_t S = (size_t)sensor_w * sensor_h;
    o->lut = (double*)malloc(S * 3 * sizeof(double));
    memcpy(o->lut, lut, S * 3 * sizeof(double));
    o->pix_start = (uint32_t*)malloc((S + 1) * sizeof(uint32_t));
    return o;
}

static void free_state(emba_oracle* o)
{
    free(o->order); free(o->pix); free(o->pm); free(o->D); free(o->cp); free(o->inl);
    free(o->polv); free(o->dp); free(o->Gpm); free(o->temp);
    o->order = NULL; o->pix = NULL; o->pm = o->D = o->dp = o->Gpm = o->temp = NULL;
    o->cp = o->inl = NULL; o->polv = NULL; o->cap = 0;
}

void emba_oracle_set_first_counted(emba_oracle* o, size_t k0) { o->first_counted = k0; }

void emba_oracle_destroy(emba_oracle* o)
{
    if (!o) return;
    free_state(o);
    free(o->lut); free(o->pix_start); free(o);
}

static void ensure_cap(emba_oracle* o, size_t n)
{
    if (n <= o->cap) return;
    free_state(o);
    o->cap = n;
    o->order = (uint32_t*)malloc(n * sizeof(uint32_t));
    o->pix = (uint32_t*)malloc(n * sizeof(uint32_t));
    o->pm = (double*)malloc(2 * n * sizeof(double));
    o->D = (double*)malloc(12 * n * sizeof(double));
    o->cp = (int32_t*)malloc(n * sizeof(int32_t));
    o->inl = (int32_t*)malloc(n * sizeof(int32_t));
    o->polv = (uint8_t*)malloc(n);
    o->dp = (double*)malloc(2 * n * sizeof(double));
    o->Gpm = (double*)malloc(2 * n * sizeof(double));
    o->temp = (double*)malloc(2 * n * sizeof(double));
}

/* a4 — EventWarper::warpEventToMap (src/utils/event_pano_warper.cpp:43-74) given R = rot.matrix() */
static void warp_with_R(const emba_oracle* o, int ev_x, int ev_y, const double* R, double* pm, double* J23)
{
    const int idx = ev_y * o->sw + ev_x;
    const double* b = o->lut + 3 * (size_t)idx;
    double rb[3];
    for (int i = 0; i < 3; ++i) rb[i] = sum3(R[3 * i] * b[0], R[3 * i + 1] * b[1], R[3 * i + 2] * b[2]);
    if (J23) {
        /* drb_ddrot = -[rb]x as written at event_pano_warper.cpp:62 */
        const double M[9] = {0, rb[2], -rb[1], -rb[2], 0, rb[0], rb[1], -rb[0], 0};
        double Jp[6];
        emba_oracle_project(o->W, o->H, rb, pm, Jp);
        /* cv::Matx23d * cv::Matx33d: s = 0; s += a(i,k)*b(k,j), k = 0..2 */
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += Jp[3 * i + k] * M[3 * k + j];
                J23[3 * i + j] = s;
            }
    } else {
        emba_oracle_project(o->W, o->H, rb, pm, NULL);
    }
}

void emba_oracle_warp(const emba_oracle* o, int ev_x, int ev_y, const double* q_xyzw, double* pm, double* J23)
{
    double R[9];
    quat_to_matrix(q_xyzw, R);
    warp_with_R(o, ev_x, ev_y, R, pm, J23);
}

/* a1 — cv::Sobel(src,dst,CV_64F,dx,dy) defaults: ksize 3, scale 1, BORDER_REFLECT_101, separable
 * [-1 0 1] (derivative axis) x [1 2 1] (smoothing axis), row pass first then column pass
 * (model.cpp:88-97; OpenCV semantics per SURVEY.md Appendix A).  [parity unpinned] */
static inline int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = (i < 0) ? -i : 2 * (n - 1) - i;
    return i;
}

static void sobel(const double* src, int H, int W, int dx, double* dst)
{
    /* dx=1: d/dx (rows: [-1 0 1], cols: [1 2 1]); dx=0: d/dy (rows: [1 2 1], cols: [-1 0 1]) */
    double* tmp = (double*)malloc((size_t)H * W * sizeof(double));
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int y = 0; y < H; ++y) {
        const double* s = src + (size_t)y * W;
        double* t = tmp + (size_t)y * W;
        for (int x = 0; x < W; ++x) {
            const double l = s[reflect101(x - 1, W)], c = s[x], r = s[reflect101(x + 1, W)];
            t[x] = dx ? (r - l) : (l + r + 2 * c);
        }
    }
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int y = 0; y < H; ++y) {
        const double* u = tmp + (size_t)reflect101(y - 1, H) * W;
        const double* c = tmp + (size_t)y * W;
        const double* d = tmp + (size_t)reflect101(y + 1, H) * W;
        double* o = dst + (size_t)y * W;
        for (int x = 0; x < W; ++x) o[x] = dx ? (u[x] + d[x] + 2 * c[x]) : (d[x] - u[x]);
    }
    free(tmp);
}

void emba_oracle_hessian(const double* Gx, const double* Gy, int H, int W, double* Gxx, double* Gxy, double* Gyy)
{
    const size_t n = (size_t)H * W;
    double* Gyx = (double*)malloc(n * sizeof(double));
    sobel(Gx, H, W, 1, Gxx);
    sobel(Gx, H, W, 0, Gxy);
    sobel(Gy, H, W, 1, Gyx);
    sobel(Gy, H, W, 0, Gyy);
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (size_t i = 0; i < n; ++i) {
        Gxx[i] = 0.125 * Gxx[i];
        Gxy[i] = 0.125 * Gxy[i];
        Gyx[i] = 0.125 * Gyx[i];
        Gyy[i] = 0.125 * Gyy[i];
        Gxy[i] = 0.5 * (Gxy[i] + Gyx[i]);
    }
    free(Gyx);
}

/* One measurement of the pairing loop, model.cpp:186-242: event kc against its predecessor kp at the same sensor pixel.
 * Returns 0 for an outlier; otherwise the residual (:221), the panorama pixel index (:209-211) and the stored dp/Gpm/temp. */
/* The index-level half of a measurement (model.cpp:194-211): displacement, outlier test, rounded panorama pixel. */
static inline int pair_index(const double* pm_c, const double* pm_p, int W, int H, double* dpx_o, double* dpy_o, size_t* pi_out)
{
    const double dpx = pm_c[0] - pm_p[0];
    const double dpy = pm_c[1] - pm_p[1];
    const double dp_norm = sqrt(dpx * dpx + dpy * dpy);
    *dpx_o = dpx; *dpy_o = dpy;
    if (dp_norm > OUTLIER_PX) return 0; /* :200-205 */
    const double rx = round(pm_c[0]), ry = round(pm_c[1]); /* :209-210 */
    /* DEFINED BEHAVIOUR where the reference is UB: a non-finite dp or a rounded
     * pixel outside [0,W)x[0,H) is an outlier (cv::Mat::at is unchecked at model.cpp:213,227). */
    if (!(dp_norm <= OUTLIER_PX) || !(rx >= 0 && rx < W && ry >= 0 && ry < H)) return 0;
    const int pm_x = (int)rx, pm_y = (int)ry;
    *pi_out = (size_t)pm_y * W + pm_x;
    return 1;
}

static inline int pair_measure(emba_oracle* o, uint32_t kc, uint32_t kp, const double* Gx, const double* Gy,
                               const double* Gxx, const double* Gxy, const double* Gyy, double* e_out, size_t* pi_out)
{
    double dpx, dpy;
    size_t pi = 0;
    const int inl = pair_index(o->pm + 2 * kc, o->pm + 2 * kp, o->W, o->H, &dpx, &dpy, &pi);
    o->dp[2 * kc] = dpx; o->dp[2 * kc + 1] = dpy;
    if (!inl) return 0;
    const double gx = Gx[pi], gy = Gy[pi];
    const double C_pred = gx * dpx + gy * dpy;               /* :217 */
    const double C_meas = 2 * (o->polv[kc] - 0.5) * o->C_th;   /* :219 */
    *e_out = C_meas - C_pred;                                /* :221 */
    *pi_out = pi;
    /* temp = Gpm + dp^T * G2pm, :233-238 */
    o->Gpm[2 * kc] = gx; o->Gpm[2 * kc + 1] = gy;
    o->temp[2 * kc] = gx + (dpx * Gxx[pi] + dpy * Gxy[pi]);
    o->temp[2 * kc + 1] = gy + (dpx * Gxy[pi] + dpy * Gyy[pi]);
    return 1;
}

/* a1-a7 — LEGM::evaluateDataError, src/emba/model.cpp:72-258 (eval_deriv = true) */
long emba_oracle_eval_data_error(emba_oracle* o, const double* knots, int K, int64_t t0_ns, int64_t dt_ns,
                                 const double* Gx, const double* Gy, const uint16_t* ex, const uint16_t* ey,
                                 const uint8_t* pol, const int64_t* t_ns, size_t n, double* ep_out,
                                 int32_t* num_ev_map, const emba_oracle_dump* dump)
{
    const int W = o->W, H = o->H;
    const size_t npix = (size_t)W * H;
    const size_t S = (size_t)o->sw * o->sh;
    const size_t num_batches = n / EVENT_BATCH; /* std::ceil of an integer division == floor (Q1), :79 */
    const size_t n_used = num_batches * EVENT_BATCH;
    ensure_cap(o, n ? n : 1);
    o->n_used = n_used;

    memset(num_ev_map, 0, npix * sizeof(int32_t)); /* :85 */

    double* Gxx = (double*)malloc(npix * sizeof(double));
    double* Gxy = (double*)malloc(npix * sizeof(double));
    double* Gyy = (double*)malloc(npix * sizeof(double));
    emba_oracle_hessian(Gx, Gy, H, W, Gxx, Gxy, Gyy); /* :88-97 */

    /* event_map_.addEvent order == stable counting sort by sensor pixel (event_map.h:34-37) */
    if (g_threads <= 1) {
        memset(o->pix_start, 0, (S + 1) * sizeof(uint32_t));
        for (size_t k = 0; k < n_used; ++k) {
            o->pix[k] = (uint32_t)ey[k] * (uint32_t)o->sw + ex[k];
            o->pix_start[o->pix[k] + 1]++;
        }
        for (size_t p = 0; p < S; ++p) o->pix_start[p + 1] += o->pix_start[p];
        uint32_t* cur = (uint32_t*)malloc(S * sizeof(uint32_t));
        memcpy(cur, o->pix_start, S * sizeof(uint32_t));
        for (size_t k = 0; k < n_used; ++k) o->order[cur[o->pix[k]]++] = (uint32_t)k;
        free(cur);
    } else {
        /* omp mode: the same stable order from per-chunk histograms (chunk c of the time-sorted events precedes chunk c+1 inside every pixel) */
        const int T = g_threads < 32 ? g_threads : 32;
        uint32_t* hist = (uint32_t*)calloc((size_t)T * S, sizeof(uint32_t));
        const size_t per = (n_used + T - 1) / T;
#pragma omp parallel for schedule(static, 1) num_threads(T)
        for (int c = 0; c < T; ++c) {
            uint32_t* h = hist + (size_t)c * S;
            const size_t k0 = per * c, k1 = (k0 + per < n_used) ? k0 + per : n_used;
            for (size_t k = k0; k < k1; ++k) {
                o->pix[k] = (uint32_t)ey[k] * (uint32_t)o->sw + ex[k];
                h[o->pix[k]]++;
            }
        }
        uint32_t run = 0;
        for (size_t p = 0; p < S; ++p) {
            o->pix_start[p] = run;
            for (int c = 0; c < T; ++c) { const uint32_t v = hist[(size_t)c * S + p]; hist[(size_t)c * S + p] = run; run += v; }
        }
        o->pix_start[S] = run;
#pragma omp parallel for schedule(static, 1) num_threads(T)
        for (int c = 0; c < T; ++c) {
            uint32_t* h = hist + (size_t)c * S;
            const size_t k0 = per * c, k1 = (k0 + per < n_used) ? k0 + per : n_used;
            for (size_t k = k0; k < k1; ++k) o->order[h[o->pix[k]]++] = (uint32_t)k;
        }
        free(hist);
    }

    /* batches :102-172 (independent of each other: in omp mode they are dealt to the threads as they stand) */
    int spline_err = 0;
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (size_t b = 0; b < num_batches; ++b) {
        const size_t bgn = EVENT_BATCH * b, end = EVENT_BATCH * (b + 1);
        const int64_t t_batch = emba_oracle_batch_mid_ns(t_ns[bgn], t_ns[end - 1]); /* :116-119 */
        double q[4], R[9], J36[18];
        int cp_idx;
        if (emba_oracle_spline_eval(knots, K, t0_ns, dt_ns, t_batch, q, R, &cp_idx, J36)) { /* :130 */
#pragma omp atomic write
            spline_err = 1;
            continue;
        }
        for (size_t k = bgn; k < end; ++k) {
            double J23[6];
            warp_with_R(o, ex[k], ey[k], R, o->pm + 2 * k, J23); /* :155 (rot.matrix() per event is identical) */
            /* dpm_ddrot (2x3) * ddrot_ddrot_cp (3x6), :156 (cv::Mat gemm, sequential k) */
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 6; ++j) {
                    double s = 0;
                    for (int kk = 0; kk < 3; ++kk) s += J23[3 * i + kk] * J36[6 * kk + j];
                    o->D[12 * k + 6 * i + j] = s;
                }
            o->polv[k] = pol[k];
            o->cp[k] = cp_idx;
            o->inl[k] = -2;
        }
    }
    if (spline_err) { free(Gxx); free(Gxy); free(Gyy); return -1; }
    for (size_t k = n_used; k < n; ++k) { o->inl[k] = -2; o->cp[k] = -1; }

    /* pairing + residual, :176-246 */
    size_t inlier_count = 0;
    if (g_threads <= 1) {
        for (size_t p = 0; p < S; ++p) {
            for (uint32_t i = o->pix_start[p] + 1; i < o->pix_start[p + 1]; ++i) {
                const uint32_t kc = o->order[i], kp = o->order[i - 1];
                double e; size_t pi;
                if (kc < o->first_counted) continue;   /* shard lead-in: inl stays -2 */
                if (!pair_measure(o, kc, kp, Gx, Gy, Gxx, Gxy, Gyy, &e, &pi)) { o->inl[kc] = -1; continue; }
                ep_out[inlier_count] = e;                                /* :221 */
                o->inl[kc] = (int32_t)inlier_count;
                inlier_count += 1;
                num_ev_map[pi] += 1;                                     /* :227 */
            }
        }
    } else {
        /* omp mode: the same measurements, numbered in the same (sensor pixel, time) order by a prefix sum over pixels */
        uint32_t* cnt = (uint32_t*)calloc(S + 1, sizeof(uint32_t));
        double* etmp = (double*)malloc((n_used ? n_used : 1) * sizeof(double));
#pragma omp parallel for schedule(dynamic, 64) num_threads(g_threads)
        for (size_t p = 0; p < S; ++p) {
            uint32_t c = 0;
            for (uint32_t i = o->pix_start[p] + 1; i < o->pix_start[p + 1]; ++i) {
                const uint32_t kc = o->order[i], kp = o->order[i - 1];
                double e; size_t pi;
                if (kc < o->first_counted) continue;
                if (!pair_measure(o, kc, kp, Gx, Gy, Gxx, Gxy, Gyy, &e, &pi)) { o->inl[kc] = -1; continue; }
                etmp[kc] = e; o->inl[kc] = 0; ++c;
#pragma omp atomic
                num_ev_map[pi] += 1;
            }
            cnt[p + 1] = c;
        }
        for (size_t p = 0; p < S; ++p) cnt[p + 1] += cnt[p];
        inlier_count = cnt[S];
#pragma omp parallel for schedule(dynamic, 64) num_threads(g_threads)
        for (size_t p = 0; p < S; ++p) {
            uint32_t idx = cnt[p];
            for (uint32_t i = o->pix_start[p] + 1; i < o->pix_start[p + 1]; ++i) {
                const uint32_t kc = o->order[i];
                if (o->inl[kc] < 0) continue;
                ep_out[idx] = etmp[kc]; o->inl[kc] = (int32_t)idx; ++idx;
            }
        }
        free(cnt); free(etmp);
    }
    free(Gxx); free(Gxy); free(Gyy);

    if (dump) {
        if (dump->pm) memcpy(dump->pm, o->pm, 2 * n_used * sizeof(double));
        if (dump->D) memcpy(dump->D, o->D, 12 * n_used * sizeof(double));
        if (dump->cp_idx) memcpy(dump->cp_idx, o->cp, n * sizeof(int32_t));
        if (dump->inlier_idx) memcpy(dump->inlier_idx, o->inl, n * sizeof(int32_t));
        if (dump->prev) {
            for (size_t k = 0; k < n; ++k) dump->prev[k] = -1;
            for (size_t p = 0; p < S; ++p)
                for (uint32_t i = o->pix_start[p] + 1; i < o->pix_start[p + 1]; ++i)
                    dump->prev[o->order[i]] = (int32_t)o->order[i - 1];
        }
        for (size_t k = 0; k < n_used; ++k) {
            const int m = o->inl[k] >= 0;
            if (dump->pm_int) {
                dump->pm_int[2 * k] = m ? (int32_t)round(o->pm[2 * k]) : -1;
                dump->pm_int[2 * k + 1] = m ? (int32_t)round(o->pm[2 * k + 1]) : -1;
            }
            if (dump->dp) { dump->dp[2 * k] = o->inl[k] != -2 ? o->dp[2 * k] : 0; dump->dp[2 * k + 1] = o->inl[k] != -2 ? o->dp[2 * k + 1] : 0; }
            if (dump->Gpm) { dump->Gpm[2 * k] = m ? o->Gpm[2 * k] : 0; dump->Gpm[2 * k + 1] = m ? o->Gpm[2 * k + 1] : 0; }
            if (dump->temp) { dump->temp[2 * k] = m ? o->temp[2 * k] : 0; dump->temp[2 * k + 1] = m ? o->temp[2 * k + 1] : 0; }
        }
    }
    return (long)inlier_count;
}

/* Lean pass (index-level results only) for flip-rate measurements at sizes where the full per-event state (177 B/event) does
 * not fit the host: pm of every event (same leaf functions as emba_oracle_eval_data_error), then the pairing rule of
 * model.cpp:179-211 applied in TIME order with a last-event-per-sensor-pixel table — the same pairs as the pixel-major walk.
 * pm_out (2n doubles) and pm_int_out (2n, -1 where the event is not an inlier measurement) may be NULL.  Returns the inlier
 * count or -1 (batch outside the spline). */
long emba_oracle_count_map(const emba_oracle* o, const double* knots, int K, int64_t t0_ns, int64_t dt_ns,
                           const uint16_t* ex, const uint16_t* ey, const int64_t* t_ns, size_t n, double* pm_out,
                           int32_t* num_ev_map, int32_t* pm_int_out)
{
    const int W = o->W, H = o->H;
    const size_t S = (size_t)o->sw * o->sh, npix = (size_t)W * H;
    const size_t num_batches = n / EVENT_BATCH, n_used = num_batches * EVENT_BATCH;
    double* pm = pm_out ? pm_out : (double*)malloc((n_used ? n_used : 1) * 2 * sizeof(double));
    memset(num_ev_map, 0, npix * sizeof(int32_t));
    int spline_err = 0;
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (size_t b = 0; b < num_batches; ++b) {
        const size_t bgn = EVENT_BATCH * b, end = EVENT_BATCH * (b + 1);
        const int64_t t_batch = emba_oracle_batch_mid_ns(t_ns[bgn], t_ns[end - 1]);
        double q[4], R[9], J36[18];
        int cp_idx;
        if (emba_oracle_spline_eval(knots, K, t0_ns, dt_ns, t_batch, q, R, &cp_idx, J36)) {
#pragma omp atomic write
            spline_err = 1;
            continue;
        }
        for (size_t k = bgn; k < end; ++k) warp_with_R(o, ex[k], ey[k], R, pm + 2 * k, NULL);
    }
    long inliers = -1;
    if (!spline_err) {
        int64_t* last = (int64_t*)malloc(S * sizeof(int64_t));
        for (size_t p = 0; p < S; ++p) last[p] = -1;
        inliers = 0;
        for (size_t k = 0; k < n_used; ++k) {
            const size_t p = (size_t)ey[k] * o->sw + ex[k];
            if (pm_int_out) { pm_int_out[2 * k] = -1; pm_int_out[2 * k + 1] = -1; }
            if (last[p] >= 0) {
                double dpx, dpy; size_t pi = 0;
                if (pair_index(pm + 2 * k, pm + 2 * (size_t)last[p], W, H, &dpx, &dpy, &pi)) {
                    num_ev_map[pi] += 1;
                    ++inliers;
                    if (pm_int_out) { pm_int_out[2 * k] = (int32_t)(pi % (size_t)W); pm_int_out[2 * k + 1] = (int32_t)(pi / (size_t)W); }
                }
            }
            last[p] = (int64_t)k;
        }
        free(last);
    }
    if (!pm_out) free(pm);
    return inliers;
}

/* One measurement of the accumulation loop of formNormalEq[IRLS], model.cpp:396-487 / :575-684. */
static inline void accumulate_measure(const emba_oracle* o, uint32_t kc, uint32_t kp, const double* ep, const int32_t* num_ev_map,
                                      const int32_t* compact, int thres, int irls, double a, int dim, double* A11, double* b1,
                                      double* A22, double* b2, double* A12, int shared)
{
    const int W = o->W;
    if (o->inl[kc] < 0) return; /* :396 */
    const int pm_x = (int)round(o->pm[2 * kc]), pm_y = (int)round(o->pm[2 * kc + 1]);
    const size_t pi = (size_t)pm_y * W + pm_x;
    if (num_ev_map[pi] < thres) return; /* :409 */
    const size_t ai = (size_t)compact[pi];
    const double ep_k = ep[o->inl[kc]]; /* :421 */
    double Yi = 1.0;
    if (irls == 2) Yi = 1.0 / (1.0 + a * ep_k * ep_k);             /* :603 */
    else if (irls == 1) { const double e = fabs(ep_k); Yi = (e < a) ? 1.0 : a / e; } /* :608-616 */
    const double ep_w = Yi * ep_k;

    const double gx = o->dp[2 * kc], gy = o->dp[2 * kc + 1]; /* dM_dGx, dM_dGy :426-427 */
    if (!shared) {
        A22[4 * ai + 0] += Yi * (gx * gx);
        A22[4 * ai + 1] += Yi * (gx * gy);
        A22[4 * ai + 2] += Yi * (gx * gy);
        A22[4 * ai + 3] += Yi * (gy * gy);
        b2[2 * ai] += gx * ep_w;
        b2[2 * ai + 1] += gy * ep_w;
    } else {   /* omp mode: a panorama pixel is fed by several threads */
#pragma omp atomic
        A22[4 * ai + 0] += Yi * (gx * gx);
#pragma omp atomic
        A22[4 * ai + 1] += Yi * (gx * gy);
#pragma omp atomic
        A22[4 * ai + 2] += Yi * (gx * gy);
#pragma omp atomic
        A22[4 * ai + 3] += Yi * (gy * gy);
#pragma omp atomic
        b2[2 * ai] += gx * ep_w;
#pragma omp atomic
        b2[2 * ai + 1] += gy * ep_w;
    }

    double jc[6], jp[6];
    const double* Dc = o->D + 12 * (size_t)kc;
    const double* Dp = o->D + 12 * (size_t)kp;
    for (int j = 0; j < 6; ++j) {
        jc[j] = o->temp[2 * kc] * Dc[j] + o->temp[2 * kc + 1] * Dc[6 + j];          /* :449 */
        jp[j] = (-o->Gpm[2 * kc]) * Dp[j] + (-o->Gpm[2 * kc + 1]) * Dp[6 + j];     /* :459 */
    }
    const size_t sc = 3 * (size_t)o->cp[kc], sp = 3 * (size_t)o->cp[kp];
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
            A11[(sc + r) + (size_t)dim * (sc + c)] += Yi * jc[r] * jc[c]; /* :454 */
        }
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
            A11[(sp + r) + (size_t)dim * (sp + c)] += Yi * jp[r] * jp[c]; /* :463 */
        }
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
            const double cross = Yi * jc[r] * jp[c];                      /* :467 */
            A11[(sc + r) + (size_t)dim * (sp + c)] += cross;              /* :468 */
            A11[(sp + c) + (size_t)dim * (sc + r)] += cross;              /* :469 */
        }
    for (int r = 0; r < 6; ++r) {
        b1[sc + r] += jc[r] * ep_w; /* :475 */
        b1[sp + r] += jp[r] * ep_w; /* :477 */
    }
    if (A12) {
        double* c0 = A12 + (size_t)dim * (2 * ai);
        double* c1 = A12 + (size_t)dim * (2 * ai + 1);
        for (int r = 0; r < 6; ++r) {
            c0[sc + r] += Yi * jc[r] * gx; c1[sc + r] += Yi * jc[r] * gy; /* :483-484 */
            c0[sp + r] += Yi * jp[r] * gx; c1[sp + r] += Yi * jp[r] * gy; /* :486-487 (dense A12: ref mode only) */
        }
    }
}

/* a8-a10 — LEGM::formNormalEq (model.cpp:316-491) / formNormalEqIRLS (:493-687) */
long emba_oracle_form_normal_eq(emba_oracle* o, const double* ep, int K, const int32_t* num_ev_map,
                                int thres, int irls, double a, double* A11, double* b1,
                                uint32_t* active_idx, double* A22, double* b2, double* A12)
{
    const int W = o->W, H = o->H;
    const size_t npix = (size_t)W * H;
    const size_t S = (size_t)o->sw * o->sh;
    const int dim = 3 * K;

    /* active set in ascending pano index (std::set order), :325-344, :371-377 */
    int32_t* compact = (int32_t*)malloc(npix * sizeof(int32_t));
    size_t P = 0;
    for (size_t i = 0; i < npix; ++i) {
        if (num_ev_map[i] >= thres) { active_idx[P] = (uint32_t)i; compact[i] = (int32_t)P; ++P; }
        else compact[i] = -1;
    }
    memset(A11, 0, (size_t)dim * dim * sizeof(double));
    memset(b1, 0, (size_t)dim * sizeof(double));
    memset(A22, 0, P * 4 * sizeof(double));
    memset(b2, 0, P * 2 * sizeof(double));
    if (A12) memset(A12, 0, (size_t)dim * 2 * P * sizeof(double));

    if (g_threads <= 1 || A12) {
        for (size_t p = 0; p < S; ++p)
            for (uint32_t i = o->pix_start[p] + 1; i < o->pix_start[p + 1]; ++i)
                accumulate_measure(o, o->order[i], o->order[i - 1], ep, num_ev_map, compact, thres, irls, a, dim, A11, b1, A22, b2, A12, 0);
    } else {
        /* omp mode: per-thread A11 / b1, merged in thread order; A22 / b2 by atomic adds */
        const int T = g_threads;
        const size_t blk = (size_t)dim * dim + dim;
        double* priv = (double*)calloc((size_t)T * blk, sizeof(double));
#pragma omp parallel num_threads(T)
        {
#ifdef _OPENMP
            const int tid = omp_get_thread_num();
#else
            const int tid = 0;
#endif
            double* a11 = priv + (size_t)tid * blk;
            double* bb1 = a11 + (size_t)dim * dim;
#pragma omp for schedule(dynamic, 64)
            for (size_t p = 0; p < S; ++p)
                for (uint32_t i = o->pix_start[p] + 1; i < o->pix_start[p + 1]; ++i)
                    accumulate_measure(o, o->order[i], o->order[i - 1], ep, num_ev_map, compact, thres, irls, a, dim, a11, bb1, A22, b2, NULL, 1);
        }
#pragma omp parallel for schedule(static) num_threads(T)
        for (size_t j = 0; j < blk; ++j) {
            double acc = 0;
            for (int t = 0; t < T; ++t) acc += priv[(size_t)t * blk + j];
            if (j < (size_t)dim * dim) A11[j] = acc; else b1[j - (size_t)dim * dim] = acc;
        }
        free(priv);
    }
    free(compact);
    return (long)P;
}

/* a11 — LEGM::applyL2Reg, model.cpp:689-719 */
void emba_oracle_apply_l2(const emba_oracle* o, size_t P, const uint32_t* active_idx, double alpha,
                          const double* Gx, const double* Gy, double* A22, double* b2)
{
    (void)o;
    for (size_t i = 0; i < P; ++i) {
        A22[4 * i + 0] += alpha;
        A22[4 * i + 3] += alpha;
        b2[2 * i] -= alpha * Gx[active_idx[i]];
        b2[2 * i + 1] -= alpha * Gy[active_idx[i]];
    }
}

/* f2 — LEGM::updateMap, model.cpp:863-903: active pixels (ascending, i-th <-> x2[2i], x2[2i+1]) are incremented (:869-878),
 * inactive pixels — all the others — are set to zero (:892-901). */
void emba_oracle_update_map(size_t P, const uint32_t* active_idx, size_t npix, const double* x2, double damping,
                            double* Gx, double* Gy)
{
    uint8_t* act = (uint8_t*)calloc(npix ? npix : 1, 1);
    for (size_t i = 0; i < P; ++i) {
        const size_t p = active_idx[i];
        Gx[p] += damping * x2[2 * i];
        Gy[p] += damping * x2[2 * i + 1];
        act[p] = 1;
    }
    for (size_t p = 0; p < npix; ++p)
        if (!act[p]) { Gx[p] = 0.0; Gy[p] = 0.0; }
    free(act);
}

/* ------------------------------------------------------------------------------------------------------------------
 * The two Eigen calls of LEGM::solveNormalEq, restated from the reference's vendored Eigen 3.3.9 and PINNED against it
 * (oracle/ref_eigen.cpp, tests/golden/eigen_solvers.npz, tests/test_oracle_pinned.py).
 * ---------------------------------------------------------------------------------------------------------------- */
/* Eigen::Matrix2d::inverse(), model.cpp:750 — Eigen/src/LU/InverseImpl.h (compute_inverse<.., 2>): invdet = 1 / determinant,
 * every entry a product with invdet; determinant = m00*m11 - m10*m01 (Eigen/src/LU/Determinant.h).  A singular block gives
 * inf / nan exactly as in the reference (no check there).  Row-major in and out. */
void emba_oracle_inverse2(const double* A, double* out)
{
    const double det = A[0] * A[3] - A[2] * A[1];
    const double invdet = 1.0 / det;
    out[0] = A[3] * invdet;
    out[2] = -A[2] * invdet;
    out[1] = -A[1] * invdet;
    out[3] = A[0] * invdet;
}

/* x = S.ldlt().solve(rhs), model.cpp:789 — Eigen::LDLT<MatrixXd, Lower>: Eigen/src/Cholesky/LDLT.h:294-395 (ldlt_inplace<Lower>::
 * unblocked, the only variant: symmetric pivoting on the largest |diagonal| entry, a zero pivot leaves its column as it is) and
 * :561-596 (_solve_impl: P, L^-1, PSEUDO-inverse of D with tolerance numeric_limits<double>::min(), L^-T, P^T).  S is n x n
 * column-major, its lower triangle is read and overwritten.  Returns what ldlt.info() would say (0 Success, 1 NumericalIssue);
 * the reference never looks at it, x is produced either way.  vecD / transp (may be NULL): D and the transpositions. */
int emba_oracle_ldlt_solve(double* A, int n, const double* rhs, double* x, double* vecD, int* transp)
{
#define M_(r, c) A[(size_t)(r) + (size_t)n * (size_t)(c)]
    int* tr = (int*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
    double* temp = (double*)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    int found_zero_pivot = 0, ret = 1;
    if (n <= 1) {                                                                   /* :305-313 */
        for (int k = 0; k < n; ++k) tr[k] = k;
    } else {
        for (int k = 0; k < n; ++k) {
            int big = k;                                                            /* :318-320: first maximum of |diag| over the tail */
            double bv = fabs(M_(k, k));
            for (int i = k + 1; i < n; ++i) if (fabs(M_(i, i)) > bv) { bv = fabs(M_(i, i)); big = i; }
            tr[k] = big;
            if (k != big) {                                                         /* :323-339: symmetric swap inside the lower triangle */
                for (int j = 0; j < k; ++j) { const double t = M_(k, j); M_(k, j) = M_(big, j); M_(big, j) = t; }
                for (int i = big + 1; i < n; ++i) { const double t = M_(i, k); M_(i, k) = M_(i, big); M_(i, big) = t; }
                { const double t = M_(k, k); M_(k, k) = M_(big, big); M_(big, big) = t; }
                for (int i = k + 1; i < big; ++i) { const double t = M_(i, k); M_(i, k) = M_(big, i); M_(big, i) = t; }
            }
            const int rs = n - k - 1;
            if (k > 0) {                                                            /* :350-356 */
                for (int j = 0; j < k; ++j) temp[j] = M_(j, j) * M_(k, j);
                double dot = 0.0;
                for (int j = 0; j < k; ++j) dot += M_(k, j) * temp[j];
                M_(k, k) -= dot;
                for (int i = k + 1; i < n; ++i) {
                    double acc = 0.0;
                    for (int j = 0; j < k; ++j) acc += M_(i, j) * temp[j];
                    M_(i, k) -= acc;
                }
            }
            const double akk = M_(k, k);
            const int pivot_is_valid = fabs(akk) > 0.0;                             /* :362-363 */
            if (k == 0 && !pivot_is_valid) {                                        /* :365-376: the whole diagonal is zero */
                for (int j = 0; j < n; ++j) {
                    tr[j] = j;
                    for (int i = j + 1; i < n; ++i) if (M_(i, j) != 0.0) ret = 0;
                }
                break;
            }
            if (rs > 0 && pivot_is_valid) { for (int i = k + 1; i < n; ++i) M_(i, k) /= akk; }   /* :378-379 */
            else if (rs > 0) { for (int i = k + 1; i < n; ++i) if (M_(i, k) != 0.0) ret = 0; }    /* :380-381 */
            if (found_zero_pivot && pivot_is_valid) ret = 0;                        /* :383-384 */
            else if (!pivot_is_valid) found_zero_pivot = 1;
        }
    }
    /* _solve_impl, :561-596 */
    for (int i = 0; i < n; ++i) x[i] = rhs[i];
    for (int k = 0; k < n; ++k) if (tr[k] != k) { const double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }        /* dst = P b */
    for (int r = 0; r < n; ++r) { double v = x[r]; for (int k = 0; k < r; ++k) v -= M_(r, k) * x[k]; x[r] = v; }  /* unit lower */
    for (int i = 0; i < n; ++i) { const double d = M_(i, i); if (fabs(d) > DBL_MIN) x[i] /= d; else x[i] = 0.0; } /* pseudo-inverse of D */
    for (int r = n - 1; r >= 0; --r) { double v = x[r]; for (int k = r + 1; k < n; ++k) v -= M_(k, r) * x[k]; x[r] = v; }
    for (int k = n - 1; k >= 0; --k) if (tr[k] != k) { const double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }   /* dst = P^T dst */
    if (vecD) for (int i = 0; i < n; ++i) vecD[i] = M_(i, i);
    if (transp) for (int i = 0; i < n; ++i) transp[i] = tr[i];
    free(tr); free(temp);
    return ret ? 0 : 1;
#undef M_
}

/* f1 — LEGM::solveNormalEq, model.cpp:721-792 */
int emba_oracle_solve_normal_eq(int n, size_t P, const double* A11, const double* A12, const double* A22, const double* b1,
                                const double* b2, double lambda, double* x1, double* x2)
{
    const size_t dm = 2 * P;
    double* Binv = (double*)malloc((P ? P : 1) * 4 * sizeof(double));
    double* W = (double*)calloc((size_t)n * (dm ? dm : 1), sizeof(double));
    double* S = (double*)malloc((size_t)n * n * sizeof(double));
    double* rhs = (double*)malloc((size_t)n * sizeof(double));
    /* A11m = A11 + lambda*diag(A11), :728-730 */
    for (int c = 0; c < n; ++c)
        for (int r = 0; r < n; ++r) S[r + (size_t)n * c] = A11[r + (size_t)n * c] + ((r == c) ? lambda * A11[r + (size_t)n * c] : 0.0);
    /* A22m_i = A22_i + lambda*diag(A22_i), inverse of each 2x2, :743-759 */
    for (size_t i = 0; i < P; ++i) {
        const double Am[4] = {A22[4 * i] + lambda * A22[4 * i], A22[4 * i + 1], A22[4 * i + 2], A22[4 * i + 3] + lambda * A22[4 * i + 3]};
        emba_oracle_inverse2(Am, Binv + 4 * i);                                              /* A22m_i.inverse(), :750 */
    }
    /* W = A12 * A22m_inv, :784 */
    for (size_t i = 0; i < P; ++i)
        for (int r = 0; r < n; ++r) {
            const double a0 = A12[r + (size_t)n * (2 * i)], a1 = A12[r + (size_t)n * (2 * i + 1)];
            W[r + (size_t)n * (2 * i)] = a0 * Binv[4 * i] + a1 * Binv[4 * i + 2];
            W[r + (size_t)n * (2 * i + 1)] = a0 * Binv[4 * i + 1] + a1 * Binv[4 * i + 3];
        }
    /* S = A11m - W * A12^T, :786 ; rhs = b1 - W*b2, :789 */
    for (int r = 0; r < n; ++r) {
        double acc = b1[r];
        for (size_t k = 0; k < dm; ++k) acc -= W[r + (size_t)n * k] * b2[k];
        rhs[r] = acc;
    }
    for (size_t k = 0; k < dm; ++k)
        for (int c = 0; c < n; ++c) {
            const double a = A12[c + (size_t)n * k];
            if (a == 0.0) continue;
            for (int r = 0; r < n; ++r) S[r + (size_t)n * c] -= W[r + (size_t)n * k] * a;
        }
    /* x1 = S.ldlt().solve(rhs), :789 */
    const int bad = emba_oracle_ldlt_solve(S, n, rhs, x1, NULL, NULL);
    /* x2 = A22m_inv * (b2 - A12^T x1), :791 */
    for (size_t i = 0; i < P; ++i) {
        double t0 = b2[2 * i], t1 = b2[2 * i + 1];
        for (int r = 0; r < n; ++r) { t0 -= A12[r + (size_t)n * (2 * i)] * x1[r]; t1 -= A12[r + (size_t)n * (2 * i + 1)] * x1[r]; }
        x2[2 * i] = Binv[4 * i] * t0 + Binv[4 * i + 1] * t1;
        x2[2 * i + 1] = Binv[4 * i + 2] * t0 + Binv[4 * i + 3] * t1;
    }
    free(Binv); free(W); free(S); free(rhs);
    return bad;
}

/* ------------------------------------------------------------------------------------------------------------------
 * The same two solvers on the SPARSE form of A12 — one rank-1 factor Yi [j_c @3cp_k ; j_p @3cp_{k-1}] (x) dp per measurement
 * (model.cpp:483-487) — for sizes where the reference's dense 3K x 2P matrix (model.cpp:358) does not fit the test host.
 * Mathematically the reference's expressions; only the order in which the sums over measurements are taken differs from the
 * dense route above (tests/test_oracle_pinned.py checks the two against each other at small sizes).
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct { uint32_t ai; int sc, sp; double jc[6], jp[6], dx, dy; } factor_t;   /* jc, jp carry the IRLS weight Yi */

/* the factors of the measurements formNormalEq[IRLS] accumulates (model.cpp:396,409), grouped by active pixel; off[P+1] */
static factor_t* build_factors(const emba_oracle* o, const double* ep, const int32_t* num_ev_map, int thres, int irls, double a,
                               size_t P, const uint32_t* active, uint32_t** off_out, size_t* M_out)
{
    const int W = o->W;
    const size_t S = (size_t)o->sw * o->sh, npix = (size_t)o->W * o->H;
    int32_t* compact = (int32_t*)malloc(npix * sizeof(int32_t));
    for (size_t i = 0; i < npix; ++i) compact[i] = -1;
    for (size_t i = 0; i < P; ++i) compact[active[i]] = (int32_t)i;
    uint32_t* off = (uint32_t*)calloc(P + 2, sizeof(uint32_t));
    for (int pass = 0; pass < 2; ++pass) {
        factor_t* F = NULL;
        uint32_t* cur = NULL;
        if (pass == 1) {
            for (size_t i = 0; i < P; ++i) off[i + 1] += off[i];
            F = (factor_t*)malloc((off[P] ? off[P] : 1) * sizeof(factor_t));
            cur = (uint32_t*)malloc((P ? P : 1) * sizeof(uint32_t));
            memcpy(cur, off, P * sizeof(uint32_t));
        }
        for (size_t p = 0; p < S; ++p)
            for (uint32_t i = o->pix_start[p] + 1; i < o->pix_start[p + 1]; ++i) {
                const uint32_t kc = o->order[i], kp = o->order[i - 1];
                if (o->inl[kc] < 0) continue;
                const int pm_x = (int)round(o->pm[2 * kc]), pm_y = (int)round(o->pm[2 * kc + 1]);
                const size_t pi = (size_t)pm_y * W + pm_x;
                if (num_ev_map[pi] < thres || compact[pi] < 0) continue;
                const uint32_t ai = (uint32_t)compact[pi];
                if (pass == 0) { off[ai + 1]++; continue; }
                const double ep_k = ep[o->inl[kc]];
                double Yi = 1.0;
                if (irls == 2) Yi = 1.0 / (1.0 + a * ep_k * ep_k);
                else if (irls == 1) { const double e = fabs(ep_k); Yi = (e < a) ? 1.0 : a / e; }
                factor_t* f = F + cur[ai]++;
                f->ai = ai; f->sc = 3 * o->cp[kc]; f->sp = 3 * o->cp[kp];
                f->dx = o->dp[2 * kc]; f->dy = o->dp[2 * kc + 1];
                const double* Dc = o->D + 12 * (size_t)kc;
                const double* Dp = o->D + 12 * (size_t)kp;
                for (int j = 0; j < 6; ++j) {
                    f->jc[j] = Yi * (o->temp[2 * kc] * Dc[j] + o->temp[2 * kc + 1] * Dc[6 + j]);
                    f->jp[j] = Yi * ((-o->Gpm[2 * kc]) * Dp[j] + (-o->Gpm[2 * kc + 1]) * Dp[6 + j]);
                }
            }
        if (pass == 1) { free(cur); free(compact); *off_out = off; *M_out = off[P]; return F; }
    }
    return NULL;
}

/* column pair (c0, c1) of A12 for one active pixel from its factors, as sparse 6-row bands written into dense scratch of length n */
static void pixel_columns(const factor_t* F, uint32_t f0, uint32_t f1, double* c0, double* c1, int* bands, int* nb_out, uint8_t* mark)
{
    int nb = 0;
    for (uint32_t f = f0; f < f1; ++f) {
        const factor_t* q = F + f;
        for (int h = 0; h < 2; ++h) {
            const int base = h ? q->sp : q->sc;
            const double* j = h ? q->jp : q->jc;
            for (int r = 0; r < 6; ++r) {
                if (!mark[base + r]) { mark[base + r] = 1; bands[nb++] = base + r; c0[base + r] = 0; c1[base + r] = 0; }
                c0[base + r] += j[r] * q->dx;
                c1[base + r] += j[r] * q->dy;
            }
        }
    }
    for (int k = 0; k < nb; ++k) mark[bands[k]] = 0;
    *nb_out = nb;
}

/* f1 on the sparse factors — LEGM::solveNormalEq, model.cpp:721-792.  A11 n x n col-major and b1 (n = 3K, UNtrimmed), A22 P x
 * [xx xy; xy yy], b2 2P: the blocks formNormalEq + applyL2Reg produced; skip = 3 drops the first control pose like
 * solver.cpp:156-165 (x1 comes back with zeros there).  Uses the state of the last evaluateDataError.  Returns 0 / 1 (zero pivot). */
int emba_oracle_solve_sparse(emba_oracle* o, const double* ep, int K, const int32_t* num_ev_map, int thres, int irls, double a,
                             const double* A11, const double* b1, size_t P, const uint32_t* active, const double* A22,
                             const double* b2, double lambda, int skip, double* x1, double* x2)
{
    const int n = 3 * K, m = n - skip;
    uint32_t* off; size_t M;
    factor_t* F = build_factors(o, ep, num_ev_map, thres, irls, a, P, active, &off, &M);
    double* S = (double*)malloc((size_t)m * m * sizeof(double));
    double* rhs = (double*)malloc((size_t)n * sizeof(double));
    double* c0 = (double*)calloc(n, sizeof(double)); double* c1 = (double*)calloc(n, sizeof(double));
    double* w0 = (double*)calloc(n, sizeof(double)); double* w1 = (double*)calloc(n, sizeof(double));
    int* bands = (int*)malloc((size_t)n * sizeof(int)); uint8_t* mark = (uint8_t*)calloc(n, 1);
    double* Binv = (double*)malloc((P ? P : 1) * 4 * sizeof(double));
    for (int c = 0; c < m; ++c)
        for (int r = 0; r < m; ++r) {
            const double v = A11[(r + skip) + (size_t)n * (c + skip)];
            S[r + (size_t)m * c] = v + ((r == c) ? lambda * v : 0.0);                       /* :728-730 */
        }
    for (int r = 0; r < n; ++r) rhs[r] = b1[r];
    for (size_t i = 0; i < P; ++i) {
        const double Am[4] = {A22[4 * i] + lambda * A22[4 * i], A22[4 * i + 1], A22[4 * i + 2], A22[4 * i + 3] + lambda * A22[4 * i + 3]};
        emba_oracle_inverse2(Am, Binv + 4 * i);                                              /* :743-759 */
        int nb;
        pixel_columns(F, off[i], off[i + 1], c0, c1, bands, &nb, mark);
        for (int k = 0; k < nb; ++k) {                                                       /* W = A12 * A22m_inv, :784 */
            const int r = bands[k];
            w0[r] = c0[r] * Binv[4 * i] + c1[r] * Binv[4 * i + 2];
            w1[r] = c0[r] * Binv[4 * i + 1] + c1[r] * Binv[4 * i + 3];
        }
        for (int k = 0; k < nb; ++k) {
            const int r = bands[k];
            rhs[r] -= w0[r] * b2[2 * i] + w1[r] * b2[2 * i + 1];                             /* b1 - W*b2, :789 */
            if (r < skip) continue;
            for (int l = 0; l < nb; ++l) {
                const int c = bands[l];
                if (c < skip) continue;
                S[(r - skip) + (size_t)m * (c - skip)] -= w0[r] * c0[c] + w1[r] * c1[c];     /* S = A11m - W*A12^T, :786 */
            }
        }
    }
    for (int r = 0; r < n; ++r) x1[r] = 0.0;
    const int bad = emba_oracle_ldlt_solve(S, m, rhs + skip, x1 + skip, NULL, NULL);         /* :789 */
    {
        for (size_t i = 0; i < P; ++i) {                                                     /* x2 = A22m_inv (b2 - A12^T x1), :791 */
            int nb;
            pixel_columns(F, off[i], off[i + 1], c0, c1, bands, &nb, mark);
            double t0 = b2[2 * i], t1 = b2[2 * i + 1];
            for (int k = 0; k < nb; ++k) { const int r = bands[k]; t0 -= c0[r] * x1[r]; t1 -= c1[r] * x1[r]; }
            x2[2 * i] = Binv[4 * i] * t0 + Binv[4 * i + 1] * t1;
            x2[2 * i + 1] = Binv[4 * i + 2] * t0 + Binv[4 * i + 3] * t1;
        }
    }
    free(F); free(off); free(S); free(rhs); free(c0); free(c1); free(w0); free(w1); free(bands); free(mark); free(Binv);
    return bad;
}

/* LEGM::solveNormalEqCG, model.cpp:794-840: Eigen::ConjugateGradient<SpMat, Lower|Upper> (default DiagonalPreconditioner,
 * max 100 iterations, tolerance 1e-6, zero initial guess) on [A11m A12; A12^T A22m] with A?m = A? + lambda*diag(A?); the
 * iteration is thirdparty/basalt-headers/thirdparty/eigen/Eigen/src/IterativeLinearSolvers/ConjugateGradient.h:28-88 restated,
 * the matrix applied through the sparse factors.  skip as above.  iters_out / err_out = cg.iterations() / cg.error(). */
int emba_oracle_solve_cg_sparse(emba_oracle* o, const double* ep, int K, const int32_t* num_ev_map, int thres, int irls, double a,
                                const double* A11, const double* b1, size_t P, const uint32_t* active, const double* A22,
                                const double* b2, double lambda, int skip, int max_iter, double tol, double* x1, double* x2,
                                int* iters_out, double* err_out)
{
    const int n = 3 * K, m = n - skip;
    const size_t N = (size_t)m + 2 * P;
    uint32_t* off; size_t M;
    factor_t* F = build_factors(o, ep, num_ev_map, thres, irls, a, P, active, &off, &M);
    double* x = (double*)calloc(N, sizeof(double)); double* r = (double*)malloc(N * sizeof(double));
    double* p = (double*)malloc(N * sizeof(double)); double* z = (double*)malloc(N * sizeof(double));
    double* t = (double*)malloc(N * sizeof(double)); double* invd = (double*)malloc(N * sizeof(double));
    double* full = (double*)calloc(n, sizeof(double));
    /* y = A v with v = [v1 (m); v2 (2P)] */
#define APPLY(v, y)                                                                                                      \
    do {                                                                                                                 \
        for (int rr = 0; rr < m; ++rr) {                                                                                 \
            double acc = 0;                                                                                              \
            for (int cc = 0; cc < m; ++cc) { const double e_ = A11[(rr + skip) + (size_t)n * (cc + skip)]; acc += (rr == cc ? e_ + lambda * e_ : e_) * (v)[cc]; } \
            (y)[rr] = acc;                                                                                               \
        }                                                                                                                \
        for (int rr = 0; rr < n; ++rr) full[rr] = 0;                                                                     \
        for (size_t i = 0; i < P; ++i) {                                                                                 \
            const double v0 = (v)[m + 2 * i], v1 = (v)[m + 2 * i + 1];                                                   \
            double s0 = 0, s1 = 0;                                                                                       \
            for (uint32_t f = off[i]; f < off[i + 1]; ++f) {                                                             \
                const factor_t* q = F + f;                                                                               \
                const double dv = q->dx * v0 + q->dy * v1;                                                               \
                double jv = 0;                                                                                           \
                for (int k = 0; k < 6; ++k) {                                                                            \
                    full[q->sc + k] += q->jc[k] * dv; full[q->sp + k] += q->jp[k] * dv;                                  \
                    if (q->sc + k >= skip) jv += q->jc[k] * (v)[q->sc + k - skip];                                       \
                    if (q->sp + k >= skip) jv += q->jp[k] * (v)[q->sp + k - skip];                                       \
                }                                                                                                        \
                s0 += q->dx * jv; s1 += q->dy * jv;                                                                      \
            }                                                                                                            \
            const double a00 = A22[4 * i], a01 = A22[4 * i + 1], a10 = A22[4 * i + 2], a11 = A22[4 * i + 3];             \
            (y)[m + 2 * i] = s0 + (a00 + lambda * a00) * v0 + a01 * v1;                                                  \
            (y)[m + 2 * i + 1] = s1 + a10 * v0 + (a11 + lambda * a11) * v1;                                              \
        }                                                                                                                \
        for (int rr = 0; rr < m; ++rr) (y)[rr] += full[rr + skip];                                                       \
    } while (0)
    for (int rr = 0; rr < m; ++rr) { const double d = A11[(rr + skip) + (size_t)n * (rr + skip)]; const double dm_ = d + lambda * d; invd[rr] = dm_ != 0 ? 1.0 / dm_ : 1.0; r[rr] = b1[rr + skip]; }
    for (size_t i = 0; i < P; ++i) {
        const double d0 = A22[4 * i] + lambda * A22[4 * i], d1 = A22[4 * i + 3] + lambda * A22[4 * i + 3];
        invd[m + 2 * i] = d0 != 0 ? 1.0 / d0 : 1.0; invd[m + 2 * i + 1] = d1 != 0 ? 1.0 / d1 : 1.0;
        r[m + 2 * i] = b2[2 * i]; r[m + 2 * i + 1] = b2[2 * i + 1];
    }
    double rhs2 = 0; for (size_t k = 0; k < N; ++k) rhs2 += r[k] * r[k];
    int it = 0; double err = 0;
    if (rhs2 != 0) {
        const double thr = fmax(tol * tol * rhs2, 2.2250738585072014e-308);
        double rn2 = rhs2;                                   /* x = 0: residual = rhs */
        if (rn2 >= thr) {
            for (size_t k = 0; k < N; ++k) p[k] = invd[k] * r[k];
            double absNew = 0; for (size_t k = 0; k < N; ++k) absNew += r[k] * p[k];
            while (it < max_iter) {
                APPLY(p, t);
                double pt = 0; for (size_t k = 0; k < N; ++k) pt += p[k] * t[k];
                const double alpha = absNew / pt;
                for (size_t k = 0; k < N; ++k) { x[k] += alpha * p[k]; r[k] -= alpha * t[k]; }
                rn2 = 0; for (size_t k = 0; k < N; ++k) rn2 += r[k] * r[k];
                if (rn2 < thr) break;
                for (size_t k = 0; k < N; ++k) z[k] = invd[k] * r[k];
                const double absOld = absNew;
                absNew = 0; for (size_t k = 0; k < N; ++k) absNew += r[k] * z[k];
                const double beta = absNew / absOld;
                for (size_t k = 0; k < N; ++k) p[k] = z[k] + beta * p[k];
                ++it;
            }
        }
        err = sqrt(rn2 / rhs2);
    }
#undef APPLY
    for (int rr = 0; rr < n; ++rr) x1[rr] = rr < skip ? 0.0 : x[rr - skip];
    for (size_t k = 0; k < 2 * P; ++k) x2[k] = x[m + k];
    if (iters_out) *iters_out = it;
    if (err_out) *err_out = err;
    free(F); free(off); free(x); free(r); free(p); free(z); free(t); free(invd); free(full);
    return 0;
}

/* a12 — 0.5*ep.dot(ep) (solver.cpp:88,265) or evaluateRobustDataCost (model.cpp:279-314) */
double emba_oracle_data_cost(const double* ep, size_t m, int irls, double a)
{
    double s = 0;
    if (irls == 0) {
        for (size_t k = 0; k < m; ++k) s += ep[k] * ep[k];
        return 0.5 * s;
    }
    if (irls == 2) {
        for (size_t k = 0; k < m; ++k) s += log1p(a * (ep[k] * ep[k]));
        return (0.5 / a) * s;
    }
    const double b = -0.5 * a * a;
    for (size_t k = 0; k < m; ++k) {
        const double e = fabs(ep[k]);
        s += (e < a) ? 0.5 * e * e : a * e + b;
    }
    return s;
}

/* a12 — 0.5*alpha*|[Gx;Gy]|^2 over ALL pixels (model.cpp:260-277; solver.cpp:90,267) */
double emba_oracle_reg_cost(const double* Gx, const double* Gy, size_t npix, double alpha)
{
    double s = 0;
    for (size_t i = 0; i < npix; ++i) s += Gx[i] * Gx[i] + Gy[i] * Gy[i];
    return 0.5 * alpha * s;
}
