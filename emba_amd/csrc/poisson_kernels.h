// emba_amd/csrc/poisson_kernels.h — SURVEY §8 f3: intensity panorama from the gradient map (gfx950, fp64 matrix cores).
//
// Reference: poisson_reconstruction::reconstructFromGradient (src/image_rec/poisson_reconstruction.cpp:9-50) +
// pde::poisolve, Dirichlet branch (src/image_rec/laplace.cpp:587-797):
//     F[i][j] = gx[i][j+1] - gx[i][j] + gy[i+1][j] - gy[i][j]   (i < H-1, j < W-1; last row and column 0)
//     R  = DST-I_2D(F) / (4 (H+1)(W+1))                          (FFTW RODFT00, unnormalised: factor 2 per dimension)
//     U  = R[i][j] / (lambda1[i] + lambda2[j]),  lambda_d[k] = -4 sin^2(pi (k+1) / (2 (n_d+1)))
//     M  = DST-I_2D(U)
// FFTW is not part of the reference tree; the transform is the published RODFT00 definition
//     Y[k] = 2 sum_j X[j] sin(pi (j+1)(k+1) / (n+1)).
// The transform lengths 2(n+1) = 2050 / 4098 have large prime factors (41, 683), so an FFT is the wrong tool here: the DST is
// applied as a dense product with the symmetric sine matrix S_n[k][j] = 2 sin(pi (j+1)(k+1)/(n+1)) — a GEMM-shaped fp64 job,
// the one place on this code path where MFMA is the bound:  M = S_H ((S_H F S_W) o C) S_W,  25.8 GFLOP at 1024 x 2048.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace emba {

typedef double pdouble4_t __attribute__((ext_vector_type(4)));

// F = d gx/dx + d gy/dy with forward differences (poisson_reconstruction.cpp:21-29); row-major H x W.
__global__ void emba_divergence_kernel(const double* __restrict__ Gx, const double* __restrict__ Gy, int H, int W, double* __restrict__ F)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)H * W) return;
    const int i = (int)(idx / W), j = (int)(idx % W);
    double f = 0.0;
    if (i < H - 1 && j < W - 1) f = Gx[idx + 1] - Gx[idx] + Gy[idx + W] - Gy[idx];   // same association as the reference
    F[idx] = f;
}

// S[k][j] = 2 sin(pi (j+1)(k+1) / (n+1)), row-major n x n (symmetric).  The argument is reduced in exact integer arithmetic
// (period 2(n+1)) and folded to [0, (n+1)/2] before the sine, so every entry is accurate to the last bits.
__global__ void emba_sine_matrix_kernel(int n, double* __restrict__ S)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n * n) return;
    const long k = idx / n, j = idx % n, m = n + 1;
    long r = ((j + 1) * (k + 1)) % (2 * m);
    double sgn = 2.0;
    if (r >= m) { r -= m; sgn = -2.0; }          // sin(x + pi) = -sin(x)
    if (2 * r > m) r = m - r;                     // sin(pi - x) = sin(x)
    S[idx] = sgn * sin(3.14159265358979323846 * (double)r / (double)m);
}

// lambda[k] = -4 sin^2(pi (k+1) / (2 (n+1)))   (laplace.cpp:700-704)
__global__ void emba_dirichlet_eigen_kernel(int n, double* __restrict__ lam)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double s = sin((3.14159265358979323846 * (double)(k + 1)) / (2.0 * (double)(n + 1)));
    lam[k] = -4.0 * (s * s);
}

// C (M x N) = A (M x K) * B (K x N), all row-major fp64, on v_mfma_f64_16x16x4_f64.
// Block tile 64 x 128 x 16, 256 threads = 4 waves in a 2 x 2 arrangement, each wave 32 x 64 = 2 x 4 MFMA tiles (32 accumulator
// doubles per lane).  A and B tiles go global -> registers -> LDS (k-major, so that the operand reads of the MFMA layout — lane l
// holds element (l&15) at k = l>>4 — are consecutive 8-B words), double-buffered: the next tile's global loads are in flight
// while the current tile's 4 k-steps x 8 MFMAs run.  12 FLOP per byte loaded from L2; MFMA-bound (fp64 matrix peak 78.6 TFLOP/s).
// epilogue: 0 plain; 1 Poisson eigen-solve  C = (acc * inv_norm) / (lam1[row] + lam2[col])   (laplace.cpp:667-731); 2 C = acc * inv_norm
struct GemmParams {
    const double* A; const double* B; double* C; int M, N, K; long lda, ldb, ldc;
    int epilogue; double inv_norm; const double* lam1; const double* lam2;
    int vec;   // lda, ldb even and A, B 16-B aligned: interior tiles use 16-B loads
    int a_kstride, a_koff;   // A's reduction index k reads column a_koff + a_kstride * k (1, 0: plain; 2, parity: every other column — the folded DST)
};

constexpr int kGemmBM = 64, kGemmBK = 16;
constexpr int kGemmLdA = kGemmBM + 16;   // k-major rows padded to 16 (mod 32) words: the 4 k-slices of an
                                                                  // MFMA operand read fall on disjoint bank halves

// BN = 128 (wave tile 32 x 64) or 64 (wave tile 32 x 32: twice the blocks, for outputs too small to give every CU two of the wide ones)
template <int kGemmBN>
__global__ __launch_bounds__(256) void emba_dgemm_kernel(GemmParams p)
{
    constexpr int kGemmLdB = kGemmBN + 16, NB = kGemmBN / 32, BQ = kGemmBN / 16;   // MFMA column tiles per wave; B doubles staged per thread
    __shared__ __attribute__((aligned(16))) double sA[2][kGemmBK * kGemmLdA];
    __shared__ __attribute__((aligned(16))) double sB[2][kGemmBK * kGemmLdB];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int el = lane & 15, kk = lane >> 4;
    // XCD-aware tile order: the 8 XCDs take contiguous runs of tile rows, so the B panels an XCD streams stay in its own L2
    const int tiles_n = (p.N + kGemmBN - 1) / kGemmBN, tiles_m = (p.M + kGemmBM - 1) / kGemmBM;
    const long n_tiles = (long)tiles_m * tiles_n;
    long bid = blockIdx.x;
    {
        const long per = (n_tiles + 7) / 8;
        bid = (bid & 7) * per + (bid >> 3);
        if (bid >= n_tiles) return;   // (whole block; before any barrier)
    }
    const int m0 = (int)(bid / tiles_n) * kGemmBM, n0 = (int)(bid % tiles_n) * kGemmBN;
    const int wm = (wv & 1) * 32, wn = (wv >> 1) * (kGemmBN / 2);

    // global -> register staging: A tile 64 x 16 (4 doubles per thread: row t/4, k (t%4)*4..+3), B tile 16 x 128 (8 per thread: k t/16, cols (t%16)*8..+7)
    const int a_row = t >> 2, a_k = (t & 3) * 4;
    const int b_k = t >> 4, b_col = (t & 15) * BQ;
    double ra[4], rb[BQ];
    auto load_tile = [&](int k0) {
        const int gr = m0 + a_row;
        if (p.vec && p.a_kstride == 1 && gr < p.M && k0 + a_k + 3 < p.K) {           // interior: 16-B loads
            const double2* q2 = reinterpret_cast<const double2*>(p.A + (size_t)p.lda * gr + k0 + a_k);
            const double2 v0 = q2[0], v1 = q2[1];
            ra[0] = v0.x; ra[1] = v0.y; ra[2] = v1.x; ra[3] = v1.y;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gk = k0 + a_k + q;
                ra[q] = (gr < p.M && gk < p.K) ? p.A[(size_t)p.lda * gr + p.a_koff + (size_t)p.a_kstride * gk] : 0.0;
            }
        }
        const int gk = k0 + b_k;
        if (p.vec && gk < p.K && n0 + b_col + BQ - 1 < p.N) {
            const double2* q2 = reinterpret_cast<const double2*>(p.B + (size_t)p.ldb * gk + n0 + b_col);
#pragma unroll
            for (int q = 0; q < BQ / 2; ++q) { const double2 v = q2[q]; rb[2 * q] = v.x; rb[2 * q + 1] = v.y; }
        } else {
#pragma unroll
            for (int q = 0; q < BQ; ++q) {
                const int gc = n0 + b_col + q;
                rb[q] = (gk < p.K && gc < p.N) ? p.B[(size_t)p.ldb * gk + gc] : 0.0;
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) sA[buf][(a_k + q) * kGemmLdA + a_row] = ra[q];
#pragma unroll
        for (int q = 0; q < BQ; ++q) sB[buf][b_k * kGemmLdB + b_col + q] = rb[q];
    };

    pdouble4_t acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = pdouble4_t{0.0, 0.0, 0.0, 0.0};

    load_tile(0);
    store_tile(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < p.K; k0 += kGemmBK) {
        const bool more = k0 + kGemmBK < p.K;
        if (more) load_tile(k0 + kGemmBK);            // in flight behind the MFMAs below
#pragma unroll
        for (int ks = 0; ks < kGemmBK; ks += 4) {
            double av[2], bv[NB];
#pragma unroll
            for (int a = 0; a < 2; ++a) av[a] = sA[buf][(ks + kk) * kGemmLdA + wm + 16 * a + el];
#pragma unroll
            for (int b = 0; b < NB; ++b) bv[b] = sB[buf][(ks + kk) * kGemmLdB + wn + 16 * b + el];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        if (more) store_tile(buf ^ 1);                // the other buffer: its readers finished before the previous barrier
        __syncthreads();
        buf ^= 1;
    }

    // C/D layout of v_mfma_f64_16x16x4_f64: lane l owns rows (l>>4) + 4r, column l&15 of each 16 x 16 tile
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm + 16 * a + kk + 4 * r, col = n0 + wn + 16 * b + el;
                if (row < p.M && col < p.N) {
                    double v = acc[a][b][r];
                    if (p.epilogue == 1) v = (v * p.inv_norm) / (p.lam1[row] + p.lam2[col]);
                    else if (p.epilogue == 2) v = v * p.inv_norm;
                    p.C[(size_t)p.ldc * row + col] = v;
                }
            }
}

// The same product with a 128 x 128 x 16 block tile and 512 threads (8 waves, 4 x 2, wave tile 32 x 64 as above): a third less operand traffic
// from L2 per flop (the 64 x 128 tile above moves 0.8 GB through L2 for the 8.6 GFLOP of one folded DST product at 2048 x 4096 — 4 TB/s at its
// 200 us).  Used when the output gives every CU a tile of this size.
constexpr int kGemmBM2 = 128, kGemmBN2 = 128;
__global__ __launch_bounds__(512) void emba_dgemm128_kernel(GemmParams p)
{
    constexpr int LdA = kGemmBM2 + 16, LdB = kGemmBN2 + 16, NB = 4;
    __shared__ __attribute__((aligned(16))) double sA[2][kGemmBK * LdA];
    __shared__ __attribute__((aligned(16))) double sB[2][kGemmBK * LdB];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int el = lane & 15, kk = lane >> 4;
    const int tiles_n = (p.N + kGemmBN2 - 1) / kGemmBN2, tiles_m = (p.M + kGemmBM2 - 1) / kGemmBM2;
    const long n_tiles = (long)tiles_m * tiles_n;
    long bid = blockIdx.x;
    {
        const long per = (n_tiles + 7) / 8;
        bid = (bid & 7) * per + (bid >> 3);
        if (bid >= n_tiles) return;
    }
    const int m0 = (int)(bid / tiles_n) * kGemmBM2, n0 = (int)(bid % tiles_n) * kGemmBN2;
    const int wm = (wv & 3) * 32, wn = (wv >> 2) * 64;
    // staging: A tile 128 x 16 (4 doubles per thread: row t/4, k (t%4)*4..+3), B tile 16 x 128 (4 per thread: k t/32, cols (t%32)*4..+3)
    const int a_row = t >> 2, a_k = (t & 3) * 4;
    const int b_k = t >> 5, b_col = (t & 31) * 4;
    double ra[4], rb[4];
    auto load_tile = [&](int k0) {
        const int gr = m0 + a_row;
        if (p.vec && p.a_kstride == 1 && gr < p.M && k0 + a_k + 3 < p.K) {
            const double2* q2 = reinterpret_cast<const double2*>(p.A + (size_t)p.lda * gr + k0 + a_k);
            const double2 v0 = q2[0], v1 = q2[1];
            ra[0] = v0.x; ra[1] = v0.y; ra[2] = v1.x; ra[3] = v1.y;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gk = k0 + a_k + q;
                ra[q] = (gr < p.M && gk < p.K) ? p.A[(size_t)p.lda * gr + p.a_koff + (size_t)p.a_kstride * gk] : 0.0;
            }
        }
        const int gk = k0 + b_k;
        if (p.vec && gk < p.K && n0 + b_col + 3 < p.N) {
            const double2* q2 = reinterpret_cast<const double2*>(p.B + (size_t)p.ldb * gk + n0 + b_col);
            const double2 v0 = q2[0], v1 = q2[1];
            rb[0] = v0.x; rb[1] = v0.y; rb[2] = v1.x; rb[3] = v1.y;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gc = n0 + b_col + q;
                rb[q] = (gk < p.K && gc < p.N) ? p.B[(size_t)p.ldb * gk + gc] : 0.0;
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) sA[buf][(a_k + q) * LdA + a_row] = ra[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) sB[buf][b_k * LdB + b_col + q] = rb[q];
    };
    pdouble4_t acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = pdouble4_t{0.0, 0.0, 0.0, 0.0};
    load_tile(0);
    store_tile(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < p.K; k0 += kGemmBK) {
        const bool more = k0 + kGemmBK < p.K;
        if (more) load_tile(k0 + kGemmBK);
#pragma unroll
        for (int ks = 0; ks < kGemmBK; ks += 4) {
            double av[2], bv[NB];
#pragma unroll
            for (int a = 0; a < 2; ++a) av[a] = sA[buf][(ks + kk) * LdA + wm + 16 * a + el];
#pragma unroll
            for (int b = 0; b < NB; ++b) bv[b] = sB[buf][(ks + kk) * LdB + wn + 16 * b + el];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm + 16 * a + kk + 4 * r, col = n0 + wn + 16 * b + el;
                if (row < p.M && col < p.N) {
                    double v = acc[a][b][r];
                    if (p.epilogue == 1) v = (v * p.inv_norm) / (p.lam1[row] + p.lam2[col]);
                    else if (p.epilogue == 2) v = v * p.inv_norm;
                    p.C[(size_t)p.ldc * row + col] = v;
                }
            }
}

// ---- Fourier analysis along ONE axis + tridiagonal solves along the other ------------------------------------------------------------
// The Dirichlet solve of laplace.cpp:587-797 is  M = S_H ((S_H F S_W) o C) S_W,  C[i][k] = 1 / (4 (H+1)(W+1) (lambda1[i] + lambda2[k])).
// S_W / sqrt(2 (W+1)) is the orthogonal eigenbasis of the 1-D second-difference operator T_W = tridiag(1, -2, 1), eigenvalues lambda2, so
//     S_W diag(1 / (lambda1[i] + lambda2)) S_W / (2 (W+1)) = (T_W + lambda1[i] I)^-1
// and the W-direction transforms are a tridiagonal solve per H-frequency i:   M = S_H X / (2 (H+1)),  X[i][:] = (T_W + lambda1[i] I)^-1 (S_H F)[i][:].
// Same solution (to rounding), a third of the arithmetic: only the two transforms along the SHORT axis remain.  Everything runs on the
// transposed planes (W x H: index j * H + i), so that the H-transforms are row-major GEMMs / contiguous vectors and the tridiagonal
// recurrences over j are coalesced over i.
__global__ __launch_bounds__(256) void emba_transpose_kernel(const double* __restrict__ src, int rows, int cols, double* __restrict__ dst)
{   // dst (cols x rows) = src (rows x cols)^T, 32 x 32 tiles through LDS
    __shared__ double tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) { const int y = by + r, x = bx + tx; if (y < rows && x < cols) tile[r][tx] = src[(size_t)y * cols + x]; }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) { const int y = bx + r, x = by + tx; if (y < cols && x < rows) dst[(size_t)y * rows + x] = tile[tx][r]; }
}

// F^T straight from the gradient maps (the divergence of emba_divergence_kernel, written transposed: one pass instead of two)
__global__ __launch_bounds__(256) void emba_divergence_T_kernel(const double* __restrict__ Gx, const double* __restrict__ Gy, int H, int W, double* __restrict__ FT)
{
    __shared__ double tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int i = by + r, j = bx + tx;
        double f = 0.0;
        if (i < H - 1 && j < W - 1) { const size_t idx = (size_t)i * W + j; f = Gx[idx + 1] - Gx[idx] + Gy[idx + W] - Gy[idx]; }   // same association as the reference
        tile[r][tx] = f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) { const int j = bx + r, i = by + tx; if (j < W && i < H) FT[(size_t)j * H + i] = tile[tx][r]; }
}

// Butterfly of the folded DST written TRANSPOSED: out (n x rows)[k][row] = (E[row][k] +- O[row][k]) * scale — the last step of the solve hands the
// panorama back in its own layout without a separate transpose pass
__global__ __launch_bounds__(256) void emba_dst_butterfly_T_kernel(const double* __restrict__ E, const double* __restrict__ O, int rows, int n, double scale,
                                                                    double* __restrict__ out)
{
    __shared__ double te[32][33], to[32][33];
    const int h = n / 2;
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // bx over k (< h), by over rows
    for (int r = ty; r < 32; r += 8) {
        const int row = by + r, k = bx + tx;
        const bool in = row < rows && k < h;
        te[r][tx] = in ? E[(size_t)row * h + k] : 0.0;
        to[r][tx] = in ? O[(size_t)row * h + k] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = bx + r, row = by + tx;
        if (k < h && row < rows) {
            const double e = te[tx][r], o = to[tx][r];
            out[(size_t)k * rows + row] = (e + o) * scale;
            out[(size_t)(n - 1 - k) * rows + row] = (e - o) * scale;
        }
    }
}

// Thomas factors of T_W + lambda1[i] I = tridiag(1, beta_i, 1), beta_i = -2 + lambda1[i] < -2 (strictly diagonally dominant: no pivoting):
// cp[j][i] = 1 / (beta_i - cp[j-1][i]), cp[0][i] = 1 / beta_i.  Once per context (like the sine matrix): W x H doubles.
__global__ void emba_thomas_coef_kernel(const double* __restrict__ lam1, int H, int W, double* __restrict__ cp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H) return;
    const double beta = -2.0 + lam1[i];
    double c = 1.0 / beta;
    cp[i] = c;
    for (int j = 1; j < W; ++j) { c = 1.0 / (beta - c); cp[(size_t)j * H + i] = c; }
}

// One sweep of the Thomas algorithm over j for every system i, in place on the transposed plane:
//   forward   y_j = cp_j (g_j - y_{j-1}),  y_{-1} = 0          (d' of the textbook)
//   backward  x_j = y_j - cp_j x_{j+1},    x_W   = 0
// Both are first-order linear recurrences  y_j = a_j y_prev + b_j  (a_j = -cp_j; b_j = cp_j g_j resp. y_j): a workgroup takes kTriSys
// systems and cuts the j range into kTriChunks chunks — (A) every thread composes its chunk's affine map, (B) one thread per system
// chains the chunks' maps, (C) every thread replays its chunk from the now known start value and writes the result.
constexpr int kTriSys = 8, kTriChunks = 128;
constexpr int kTriMaxQ = 32;    // elements per chunk the recomputing backward replay supports (W <= kTriChunks * kTriMaxQ = 4096; wider planes read the table)
// The Thomas factors are RECOMPUTED inside a chunk from the chunk's first stored one (c_j = 1 / (beta_i - c_{j-1}): one division per element,
// on an otherwise idle VALU) instead of being streamed: a sweep read cp and the plane twice each — 335 MB at 2048 x 4096 — now 67 x 2 + 67 MB
// of plane traffic and 128 factors per system.  (Same values: the table itself was filled by this recurrence, emba_thomas_coef_kernel.)
template <bool BACKWARD>
__global__ __launch_bounds__(kTriSys * kTriChunks) void emba_tridiag_sweep_kernel(const double* __restrict__ cp, const double* __restrict__ lam1,
                                                                                  double* __restrict__ v, int H, int W)
{
    __shared__ double sA[kTriChunks][kTriSys], sB[kTriChunks][kTriSys], sY[kTriChunks][kTriSys];
    const int s = threadIdx.x % kTriSys, ch = threadIdx.x / kTriSys;
    const int i = blockIdx.x * kTriSys + s;
    const int q = (W + kTriChunks - 1) / kTriChunks;
    const int j0 = ch * q, j1 = (j0 + q < W) ? j0 + q : W;
    const bool live = i < H && j0 < W;
    const double beta = live ? -2.0 + lam1[i] : -3.0;
    // c_{j0-1} (0 in front of the first element: c_0 = 1 / beta), then c_j for the chunk by the recurrence — forwards in both sweeps
    const double c_in = (live && j0 > 0) ? cp[(size_t)(j0 - 1) * H + i] : 0.0;
    double A = 1.0, B = 0.0;
    if (live) {
        if (!BACKWARD) {
            double c = c_in;
            for (int j = j0; j < j1; ++j) { c = 1.0 / (beta - c); const double g = v[(size_t)j * H + i]; const double a = -c, b = c * g; B = a * B + b; A = a * A; }
        } else {
            // x_j = y_j - c_j x_{j+1}, j descending: the chunk's affine map in terms of x entering from above needs c_j in DESCENDING order; the
            // composition is associative, so build it ascending instead:  (x_{j1} -> x_{j0})  =  f_{j0} o f_{j0+1} o ... o f_{j1-1},  f_j(x) = g_j - c_j x
            double c = c_in;
            for (int j = j0; j < j1; ++j) { c = 1.0 / (beta - c); const double g = v[(size_t)j * H + i]; B = B + A * g; A = A * (-c); }
        }
    }
    sA[ch][s] = A; sB[ch][s] = B;
    __syncthreads();
    if (ch == 0) {      // chain the chunks (in sweep order); sY[c] = value entering chunk c
        double y = 0.0;
        if (!BACKWARD) for (int c = 0; c < kTriChunks; ++c) { sY[c][s] = y; y = sA[c][s] * y + sB[c][s]; }
        else for (int c = kTriChunks - 1; c >= 0; --c) { sY[c][s] = y; y = sA[c][s] * y + sB[c][s]; }
    }
    __syncthreads();
    if (!live) return;
    double y = sY[ch][s];
    if (!BACKWARD) {
        double c = c_in;
        for (int j = j0; j < j1; ++j) { c = 1.0 / (beta - c); const double g = v[(size_t)j * H + i]; y = c * (g - y); v[(size_t)j * H + i] = y; }
    } else {
        // descending replay needs c_j descending: the chunk's factors are kept from one ascending pass (q <= kTriMaxQ of them; longer chunks read the table)
        if (q <= kTriMaxQ) {
            double cs[kTriMaxQ];
            double c = c_in;
#pragma unroll
            for (int k = 0; k < kTriMaxQ; ++k) { if (j0 + k < j1) { c = 1.0 / (beta - c); cs[k] = c; } }
#pragma unroll
            for (int k = kTriMaxQ - 1; k >= 0; --k) { if (j0 + k < j1) { const int j = j0 + k; const double g = v[(size_t)j * H + i]; y = g - cs[k] * y; v[(size_t)j * H + i] = y; } }
        } else {
            for (int j = j1 - 1; j >= j0; --j) { const double c = cp[(size_t)j * H + i], g = v[(size_t)j * H + i]; y = g - c * y; v[(size_t)j * H + i] = y; }
        }
    }
}

// ---- the sine matrix folded by its symmetry  S[n-1-k][j] = (-1)^j S[k][j]  (n even) ----------------------------------------------------
//     y_k + y_{n-1-k} = 2 sum_{j even} S[k][j] x_j =: 2 E_k ,   y_k - y_{n-1-k} = 2 sum_{j odd} S[k][j] x_j =: 2 O_k      (k < n/2)
// so a DST-I of length n is two (n/2 x n/2) products on the even- and odd-indexed inputs and a butterfly: half the multiplications.
// Sf[parity][jp][k] = S[k][2 jp + parity] = 2 sin(pi (2 jp + parity + 1)(k + 1) / (n + 1)),  jp, k < n/2  (row-major, the B operand of the GEMM)
__global__ void emba_sine_folded_kernel(int n, double* __restrict__ Sf)
{
    const long h = n / 2, idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * h * h) return;
    const long parity = idx / (h * h), rem = idx % (h * h), jp = rem / h, k = rem % h, m = n + 1;
    long r = ((2 * jp + parity + 1) * (k + 1)) % (2 * m);
    double sgn = 2.0;
    if (r >= m) { r -= m; sgn = -2.0; }
    if (2 * r > m) r = m - r;
    Sf[idx] = sgn * sin(3.14159265358979323846 * (double)r / (double)m);
}
// y[row][k] = (E + O) * scale, y[row][n-1-k] = (E - O) * scale from EO = [E (rows x n/2) | O (rows x n/2)] stored as two planes
__global__ void emba_dst_butterfly_kernel(const double* __restrict__ E, const double* __restrict__ O, int rows, int n, double scale, double* __restrict__ y)
{
    const long h = n / 2, idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)rows * h) return;
    const long row = idx / h, k = idx % h;
    const double e = E[idx], o = O[idx];
    y[row * n + k] = (e + o) * scale;
    y[row * n + (n - 1 - k)] = (e - o) * scale;
}

}  // namespace emba
