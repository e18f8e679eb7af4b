// emba_amd/csrc/solve_kernels.h — SURVEY §8f1: the Schur-complement solve of LEGM::solveNormalEq (model.cpp:721-792) on the
// device, from the SPARSE A12 factors (records) instead of the reference's dense 3K x 2P matrix.
//
//   A11m = A11 + lambda*diag(A11)                              model.cpp:728-730
//   A22m_i = A22_i + lambda*diag(A22_i),  B_i = A22m_i^-1      :746-750
//   S  = A11m - A12 * B * A12^T                                :784-786
//   x1 = S \ (b1 - A12*B*b2)                                   :789
//   x2 = B * (b2 - A12^T * x1)                                 :791
//
// With the 2x2 Cholesky factor A22m_i = C_i C_i^T the products factor as U = A12 * C^-T (per pixel a 3K x 2 column pair),
// y = C^-1 b2:   S = A11m - U U^T,  rhs = b1 - U y,  x2 = C^-T (y - U^T x1).  U is built densely a chunk of pixels at a time
// (one wave per pixel, its two columns assembled in LDS from the records of that pixel — no global atomics), the two big
// products are plain library SYRK/GEMV calls (rocBLAS), the 3K x 3K factorization is a single-workgroup Cholesky.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace emba {

// ---- per-pixel record lists (CSR over the active pixels, compact order) ------------------------------------------------
// level 1: per-block exclusive scan of cnt[i] = count[active_idx[i]]
__global__ __launch_bounds__(256) void emba_csr_scan1_kernel(const int32_t* __restrict__ count, const uint32_t* __restrict__ active_idx,
                                                             long P, uint32_t* __restrict__ off, uint32_t* __restrict__ blk_tot)
{
    __shared__ uint32_t s_w[4];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const long i0 = ((long)blockIdx.x * 256 + t) * 8;
    uint32_t v[8], mine = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = (i0 + k < P) ? (uint32_t)count[active_idx[i0 + k]] : 0u; mine += v[k]; }
    uint32_t x = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_w[wv] = x;
    __syncthreads();
    uint32_t run = x - mine;
    for (int w = 0; w < wv; ++w) run += s_w[w];
#pragma unroll
    for (int k = 0; k < 8; ++k) { if (i0 + k < P) off[i0 + k] = run; run += v[k]; }
    if (t == 255) blk_tot[blockIdx.x] = run;
}

// level 3: add the scanned block totals; off[P] = total
__global__ void emba_csr_scan3_kernel(uint32_t* __restrict__ off, long P, const uint32_t* __restrict__ blk_off, const uint32_t* __restrict__ total)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) off[i] += blk_off[i / 2048];
    if (i == 0) off[P] = total[0];
}

// bucket fill: every active record takes a ticket in its pixel's list
__global__ void emba_csr_fill_kernel(const double* __restrict__ rec, long n_slots, const int32_t* __restrict__ count,
                                     const int32_t* __restrict__ compact, int thres, const uint32_t* __restrict__ off,
                                     uint32_t* __restrict__ cursor, uint32_t* __restrict__ bucket)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    const double2 tail = reinterpret_cast<const double2*>(rec + (size_t)kRecStride * s)[7];
    const uint32_t pi = (uint32_t)__double2loint(tail.y);
    if (pi == kInvalidPix || count[pi] < thres) return;
    const int32_t k = compact[pi];
    const uint32_t pos = off[k] + atomicAdd(cursor + k, 1u);
    bucket[pos] = (uint32_t)s;
}

// ---- U chunk: one wave per active pixel ---------------------------------------------------------------------------------
struct SchurBuildParams {
    const double* rec; const uint32_t* slot_key; const uint32_t* off; const uint32_t* bucket;
    const double* A22b2; long p0, p1;            // pixel chunk [p0, p1) in compact order
    double lambda; int irls; double eta; int n;  // n = 3K
    double* U; long ldu;                         // column-major n x 2(p1-p0)
    double* yv; double* cfac;                    // per pixel: y = C^-1 b2 (2), C = {c00, c10, c11}
    int* info;                                   // set to 1 if some A22m is not positive definite
};

__global__ __launch_bounds__(256) void emba_schur_build_kernel(SchurBuildParams p)
{
    extern __shared__ __attribute__((aligned(16))) double s_cols[];   // 4 waves x 2 columns x n
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* c0 = s_cols + (size_t)wv * 2 * p.n;
    double* c1 = c0 + p.n;
    const long nwaves = (long)gridDim.x * 4;
    for (long i = p.p0 + (long)blockIdx.x * 4 + wv; i < p.p1; i += nwaves) {
        const double* q = p.A22b2 + 5 * i;
        const double mxx = q[0] + p.lambda * q[0], mxy = q[1], myy = q[2] + p.lambda * q[2];   // model.cpp:748
        const double c00 = sqrt(mxx), c10 = mxy / c00, c11 = sqrt(myy - c10 * c10);
        if (!(mxx > 0.0) || !(myy - c10 * c10 > 0.0)) { if (lane == 0) atomicOr(p.info, 1); }
        const double y0 = q[3] / c00, y1 = (q[4] - c10 * y0) / c11;
        if (lane == 0) { p.yv[2 * i] = y0; p.yv[2 * i + 1] = y1; p.cfac[3 * i] = c00; p.cfac[3 * i + 1] = c10; p.cfac[3 * i + 2] = c11; }
        for (int r = lane; r < p.n; r += 64) { c0[r] = 0.0; c1[r] = 0.0; }
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint32_t b0 = p.off[i], b1 = p.off[i + 1];
        for (uint32_t b = b0; b < b1; ++b) {
            const uint32_t s = p.bucket[b];
            const double x = (lane < 16) ? p.rec[(size_t)kRecStride * s + lane] : 0.0;
            const double dx = __shfl(x, 12), dy = __shfl(x, 13), e = __shfl(x, 14);
            double w = 1.0;
            if (p.irls == 2) w = 1.0 / (1.0 + p.eta * e * e);
            else if (p.irls == 1) { const double a = fabs(e); w = (a < p.eta) ? 1.0 : p.eta / a; }
            const uint32_t key = p.slot_key[s];
            const int bc = 3 * (int)(key >> 16), bp = 3 * (int)(key & 0xFFFFu);
            const double wx = w * x;                                            // Yi_inv * dM_ddrot^T, model.cpp:483-487 / 679-683
            if (lane < 6) { c0[bc + lane] += wx * dx; c1[bc + lane] += wx * dy; }
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // rows of c and p may overlap: two ordered steps
            if (lane >= 6 && lane < 12) { c0[bp + lane - 6] += wx * dx; c1[bp + lane - 6] += wx * dy; }
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // U = A12 * C^-T :  u0 = a0/c00 ;  u1 = (a1 - a0*c10/c00)/c11
        double* u0 = p.U + (size_t)p.ldu * (2 * (i - p.p0));
        double* u1 = u0 + p.ldu;
        for (int r = lane; r < p.n; r += 64) {
            const double a0 = c0[r], a1 = c1[r];
            const double t0 = a0 / c00;
            u0[r] = t0;
            u1[r] = (a1 - t0 * c10) / c11;
        }
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// S = A11 + lambda*diag(A11) (full square copy; SYRK then updates the lower triangle), rhs = b1
__global__ void emba_schur_init_kernel(const double* __restrict__ A11, const double* __restrict__ b1, int n, double lambda,
                                       double* __restrict__ S, double* __restrict__ rhs)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)n * n) {
        const int r = (int)(i % n), c = (int)(i / n);
        const double a = A11[i];
        S[i] = (r == c) ? a + lambda * a : a;       // model.cpp:728-730
    }
    if (i < n) rhs[i] = b1[i];
}

// In-place Cholesky (lower, column-major, leading dimension ld) of the m x m matrix A and solution of A x = b (b overwritten by x).
// One workgroup.  info[0] |= 2 if a pivot is not positive.
__global__ __launch_bounds__(1024) void emba_chol_solve_kernel(double* __restrict__ A, int m, int ld, double* __restrict__ b, int* __restrict__ info)
{
    __shared__ double s_piv;
    const int t = threadIdx.x, nt = blockDim.x;
    for (int j = 0; j < m; ++j) {
        if (t == 0) {
            const double d = A[(size_t)j * ld + j];
            if (!(d > 0.0)) atomicOr(info, 2);
            s_piv = sqrt(d);
        }
        __syncthreads();
        const double piv = s_piv;
        for (int r = j + t; r < m; r += nt) A[(size_t)j * ld + r] = (r == j) ? piv : A[(size_t)j * ld + r] / piv;
        __syncthreads();
        // trailing update of the lower triangle: A[r][c] -= L[r][j] * L[c][j]  for j < c <= r
        const int rem = m - j - 1;
        for (long idx = t; idx < (long)rem * rem; idx += nt) {
            const int c = j + 1 + (int)(idx / rem), r = j + 1 + (int)(idx % rem);
            if (r >= c) A[(size_t)c * ld + r] -= A[(size_t)j * ld + r] * A[(size_t)j * ld + c];
        }
        __syncthreads();
    }
    // forward substitution L z = b, then back substitution L^T x = z (column-oriented, one column per step)
    for (int j = 0; j < m; ++j) {
        if (t == 0) b[j] /= A[(size_t)j * ld + j];
        __syncthreads();
        const double bj = b[j];
        for (int r = j + 1 + t; r < m; r += nt) b[r] -= A[(size_t)j * ld + r] * bj;
        __syncthreads();
    }
    for (int j = m - 1; j >= 0; --j) {
        if (t == 0) b[j] /= A[(size_t)j * ld + j];
        __syncthreads();
        const double bj = b[j];
        for (int r = t; r < j; r += nt) b[r] -= A[(size_t)r * ld + j] * bj;   // L^T[r][j] = L[j][r], stored at column r, row j
        __syncthreads();
    }
}

// x2_i = C_i^-T (y_i - z_i)     with z = U^T x1                                     model.cpp:791
__global__ void emba_schur_x2_kernel(const double* __restrict__ yv, const double* __restrict__ z, const double* __restrict__ cfac,
                                     long p0, long p1, double* __restrict__ x2)
{
    const long i = p0 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p1) return;
    const double t0 = yv[2 * i] - z[2 * (i - p0)], t1 = yv[2 * i + 1] - z[2 * (i - p0) + 1];
    const double c00 = cfac[3 * i], c10 = cfac[3 * i + 1], c11 = cfac[3 * i + 2];
    const double b = t1 / c11;
    x2[2 * i + 1] = b;
    x2[2 * i] = (t0 - c10 * b) / c00;
}

}  // namespace emba
