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
// (one wave per pixel, its two columns assembled in LDS from the records of that pixel — no global atomics) with y as an extra
// ROW, so that one split-K SYRK on the fp64 matrix cores (v_mfma_f64_16x16x4_f64; n is small, the K dimension is 2P: library
// SYRK/GEMM kernels do not split K and take 10 ms here) yields S and rhs together; the n x n factorization is a blocked
// right-looking Cholesky whose trailing updates reuse the same SYRK kernel; x2 comes straight from the records.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include "kernels.h"

namespace emba {

// ---- per-pixel record lists (CSR over the active pixels, compact order) ------------------------------------------------
// Two views of "the records of this solve":
//   local  : the context's own records (slot order); a record takes part iff its stamp is the current evaluation's and its pixel is
//            in the active set (compact[pano] >= 0: the active set may come from all-reduced counts, so the count map is NOT consulted
//            and the list lengths are COUNTED from the records — a pixel can be active without a single local record)
//   packed : records a sharded solve has re-distributed by pixel owner (emba_solve_shard_pack): all valid, tail word = {compact pixel
//            index, control-pose pair key}
struct RecView {
    const double* rec; const uint32_t* slot_key; const int32_t* compact; uint32_t stamp; int packed; long pix_base;   // pix_base: first compact index of this rank's pixels (packed view)
};

__device__ __forceinline__ bool rec_pixel(const RecView& v, long s, int32_t& k)
{
    const double tail = v.rec[(size_t)kRecStride * s + 15];
    if (v.packed) { k = (int32_t)((long)(uint32_t)__double2loint(tail) - v.pix_base); return true; }
    uint32_t pi;
    if (!record_valid(tail, v.stamp, pi)) return false;
    k = v.compact[pi];
    return k >= 0;
}
__device__ __forceinline__ uint32_t rec_key(const RecView& v, uint32_t s)
{
    return v.packed ? (uint32_t)__double2hiint(v.rec[(size_t)kRecStride * s + 15]) : v.slot_key[s];
}

// compile-time loop: f(std::integral_constant<int, B>) ... f(std::integral_constant<int, E - 1>)
template <int B, int E, class F> __device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// DPP row rotation of a double inside its row of 16 lanes (VALU moves, no LDS crossbar)
template <int N> __device__ __forceinline__ double dpp_row_ror(double v)      // lane l <- lane (l - N) mod 16 of its row
{
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + N, 0xf, 0xf, true),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + N, 0xf, 0xf, true));
}

__global__ void emba_csr_count_kernel(RecView v, long n_rec, uint32_t* __restrict__ cnt)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_rec) return;
    int32_t k;
    if (rec_pixel(v, s, k)) atomicAdd(cnt + k, 1u);
}

// The same counts without a pass over the records: a valid record with pixel p <=> an inlier measurement counted at p by the evaluation that wrote it
// (the warp kernels write the record and count the measurement together), so while the context's OWN count map still holds the materialised counts of that
// evaluation (emba_ctx::count_stamp) the list length of active pixel k is count[active_idx[k]].  (csr_count: 250 us at config 2's shape — it fetches every
// record's line for its tail word.)
__global__ void emba_csr_count_from_map_kernel(const uint32_t* __restrict__ active_idx, const int32_t* __restrict__ count, long P, uint32_t* __restrict__ cnt)
{
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < P) { const int32_t v = count[active_idx[k]]; cnt[k] = v > 0 ? (uint32_t)v : 0u; }
}

// Sorted fill: every participating record takes a ticket in its pixel's list and is COPIED there (eight lanes per 128-B record: coalesced
// reads in slot order, one full line written per record), with the tail rewritten to the packed form {pixel of the list, pair key}.  The
// U build, the x2 kernel and every iteration of the CG solver then stream a pixel's records from consecutive lines instead of gathering
// random 128-B lines through an index list (config 2's shape: 7.5 M records, each pass ran at the ~2 TB/s of that gather).
// (round 6, measured and dropped: one ticket per RUN of same-pixel records — a wave reads 64 tail words, the head lane of a run adds its length — instead of a
// returning atomic per record: 457 -> 556 us at config 2's shape; the kernel is bound by its copy, not by the tickets.  profiles/r06_schur_pipe_ab.txt)
__global__ __launch_bounds__(256) void emba_csr_fill_sorted_kernel(RecView v, long n_rec, const uint32_t* __restrict__ off, uint32_t* __restrict__ cursor,
                                                                   double* __restrict__ out)
{
    const long s = (long)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int lane = threadIdx.x & 63, c8 = threadIdx.x & 7;
    const long sc = s < n_rec ? s : n_rec - 1;                       // (n_rec >= 1 whenever this is launched; every lane stays for the shuffles)
    const double2 val = reinterpret_cast<const double2*>(v.rec + (size_t)kRecStride * sc)[c8];
    int32_t k = -1; uint32_t pos = 0, key = 0;
    if (c8 == 7 && s < n_rec) {                                      // the lane that holds the tail word
        bool ok;
        if (v.packed) { k = (int32_t)((long)(uint32_t)__double2loint(val.y) - v.pix_base); ok = true; }
        else { uint32_t pi; ok = record_valid(val.y, v.stamp, pi); if (ok) { k = v.compact[pi]; ok = k >= 0; } }
        if (ok) { pos = off[k] + atomicAdd(cursor + k, 1u); key = v.packed ? (uint32_t)__double2hiint(val.y) : v.slot_key[s]; if (pos >= (uint32_t)n_rec) k = -1; }   // (never, with true counts: no write past the buffer whatever the counts)
        else k = -1;
    }
    k = __shfl(k, lane | 7); pos = (uint32_t)__shfl((int)pos, lane | 7); key = (uint32_t)__shfl((int)key, lane | 7);
    if (k < 0) return;
    double2 o = val;
    if (c8 == 7) o.y = __hiloint2double((int)key, k);
    reinterpret_cast<double2*>(out + (size_t)kRecStride * pos)[c8] = o;
}

// 1 / sqrt(d) for a positive, normal d: v_rsq_f64 (about 26 good bits) + two Newton steps.  The factorisation multiplies by it instead of
// dividing by the square root: the pivot chain of a 64-column panel is 64 x (sqrt + divide) ~ 64 x 450 cycles otherwise, most of the kernel.
__device__ __forceinline__ double rsqrt_nr(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    y = y * fma(-h, y * y, 1.5);
    y = y * fma(-h, y * y, 1.5);
    return y;
}

// ---- U chunk: one wave per active pixel ---------------------------------------------------------------------------------
// Column slices of the block-sparse SYRK: kSyrkSlicePix consecutive active pixels (ascending panorama index = a piece of a panorama row).
// The control poses a pixel's measurements involve are those in view while the camera looks at it, so over a long window (config 2:
// 10 s, K = 201) the columns of a slice are non-zero in a BAND of rows only and most (64-row block pair, slice) products vanish.
#ifndef SYRK_SLICE_PIX
#define SYRK_SLICE_PIX 128
#endif
constexpr int kSyrkSlicePix = SYRK_SLICE_PIX;

struct SchurBuildParams {
    RecView view; const uint32_t* off;            // view: the records in pixel order (build_lists); off: first record of every pixel
    const double* A22b2; long p0, p1;            // pixel chunk [p0, p1) in compact order
    double lambda; int irls; double eta; int n;  // n = 3K
    double* U; long ldu;                         // column-major n x 2(p1-p0)
    double* yv; double* cfac;                    // per pixel: y = C^-1 b2 (2), C = {c00, c10, c11}
    int* info;                                   // set to 1 if some A22m is not positive definite
    unsigned long long* slice_mask;              // per slice of kSyrkSlicePix pixels of the chunk: which 64-row blocks of U its columns touch (nullptr: not wanted)
    uint16_t* range;                             // per pixel of the chunk: lo | hi << 8, the 16-ROW groups [lo, hi] its two columns of U are non-zero in (lo > hi: none);
                                                 // the columns are WRITTEN in the 64-row blocks [lo >> 2, hi >> 2]
    double* rhs_row; long lds;                   // rhs_row[lds * r] -= (U y)[r]: row n of the augmented S (the right-hand side b1 - U y), accumulated here
    const uint32_t* perm;                        // column order of U: position j holds compact pixel perm[j] (nullptr: j).  off / A22b2 / yv / cfac stay indexed by the pixel
};

#ifndef SCHUR_BUILD_WAVES
#define SCHUR_BUILD_WAVES 4
#endif
constexpr int kBuildWaves = SCHUR_BUILD_WAVES;     // waves per workgroup of the U build: (2 kBuildWaves + 1) n doubles of LDS.  Measured at config 2's shape: 3 waves 943 us,
                                                   // 4: 906, 5: 1643, 6: 1439 (five would fit three workgroups = 15 waves per CU in LDS, and is slower all the same)
__global__ __launch_bounds__(64 * kBuildWaves) void emba_schur_build_kernel(SchurBuildParams p)
{
    extern __shared__ __attribute__((aligned(16))) double s_cols[];   // 4 waves x 2 columns x n, then the block's n partial sums of U y
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* c0 = s_cols + (size_t)wv * 2 * p.n;
    double* c1 = c0 + p.n;
    double* s_rhs = s_cols + (size_t)kBuildWaves * 2 * p.n;
    for (int r = threadIdx.x; r < (2 * kBuildWaves + 1) * p.n; r += 64 * kBuildWaves) s_cols[r] = 0.0;   // (the columns are re-zeroed after every pixel, where they were touched)
    __syncthreads();
    const long nwaves = (long)gridDim.x * kBuildWaves;
    // Software pipeline over the wave's pixels: the list bounds and 2x2 block of pixel i+1 are fetched while pixel i is worked on; the pixel's
    // records are consecutive (emba_csr_fill_sorted_kernel), so a pixel costs ONE dependent round trip.
    const long i_first = p.p0 + (long)blockIdx.x * kBuildWaves + wv, i_last = p.p1 - 1;       // (loads of pixels past the end are clamped, never used)
    struct Hdr { uint32_t b0, b1; long k; double q0, q1, q2, q3, q4; };
    auto load_hdr = [&](long i, Hdr& h) {
        const long ic0 = i < i_last ? i : i_last;
        const long ic = p.perm ? (long)p.perm[ic0] : ic0;            // the compact pixel whose two columns sit at position i
        h.k = ic;
        h.b0 = p.off[ic]; h.b1 = p.off[ic + 1];
        const double* q = p.A22b2 + 5 * ic;
        h.q0 = q[0]; h.q1 = q[1]; h.q2 = q[2]; h.q3 = q[3]; h.q4 = q[4];
    };
    // 16 records in flight at a time (lane l: element l&15 of record l>>4 of each group of four)
    struct Grp { double x[4]; double2 dxy[4], et[4]; };
    const int el = lane & 15, kk = lane >> 4;
    auto load_grp = [&](uint32_t base, int m, int t0, Grp& g) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = t0 + 4 * u + kk;
            const double* rp = p.view.rec + (size_t)kRecStride * (base + (uint32_t)(r < m ? r : 0));
            g.x[u] = rp[el];
            // dp, the residual and the tail word straight from the record's line (same address in the record's 16 lanes: one cached
            // request) rather than broadcast from lanes 12-14 through the LDS crossbar, which this kernel's atomics already load
            g.dxy[u] = reinterpret_cast<const double2*>(rp)[6];
            g.et[u] = reinterpret_cast<const double2*>(rp)[7];
        }
    };
    // (round 4) the FIRST record group of pixel i+1 is fetched while pixel i is worked on, its list bounds one pixel earlier still: a pixel used to
    // cost one dependent round trip to its records with nothing else of the wave in flight (three waves per SIMD at K = 201: LDS).  TWO pixels ahead was
    // measured too: 216 VGPRs, two waves per SIMD, 1024 vs 907 us at config 2's shape, 100 vs 78 at K = 21
    Hdr h_cur, h_nxt, h_nn;
    Grp g_cur, g_nxt;
    load_hdr(i_first, h_cur);
    load_hdr(i_first + nwaves, h_nxt);
    load_grp(h_cur.b0, (int)(h_cur.b1 - h_cur.b0), 0, g_cur);
    for (long i = i_first; i < p.p1; i += nwaves) {
        load_hdr(i + 2 * nwaves, h_nn);
        load_grp(h_nxt.b0, (int)(h_nxt.b1 - h_nxt.b0), 0, g_nxt);
        const double mxx = h_cur.q0 + p.lambda * h_cur.q0, mxy = h_cur.q1, myy = h_cur.q2 + p.lambda * h_cur.q2;   // model.cpp:748
        // 2x2 Cholesky by reciprocal square roots (v_rsq_f64 + two Newton steps, rsqrt_nr): two square roots, three divisions and two reciprocals per pixel were
        // ~150 fp64 VALU instructions that every lane of the wave executed (round 5); the factor agrees with sqrt / divide to an ulp or two (x2 is compared at 1e-7)
        const double rs0 = rsqrt_nr(mxx);
        const double c00 = mxx * rs0, c10 = mxy * rs0, dd = myy - c10 * c10;
        const double rs1 = rsqrt_nr(dd);
        const double c11 = dd * rs1;
        if (!(mxx > 0.0) || !(dd > 0.0)) { if (lane == 0) atomicOr(p.info, 1); }
        const double y0 = h_cur.q3 * rs0, y1 = (h_cur.q4 - c10 * y0) * rs1;
        if (lane == 0) { const long k = h_cur.k; p.yv[2 * k] = y0; p.yv[2 * k + 1] = y1; p.cfac[3 * k] = c00; p.cfac[3 * k + 1] = c10; p.cfac[3 * k + 2] = c11; }
        const uint32_t b0 = h_cur.b0, b1 = h_cur.b1;
        unsigned long long rows_mask = 0ull;
        int rmin = 0x7FFFFFFF, rmax = -1;                               // first / last row of U the pixel's records touch
        // The sums go to the LDS columns with fp64 LDS atomics: rows of different records, or of a record's c and p halves, may coincide.
        {
            const uint32_t base = b0; const int m = (int)(b1 - b0);
            for (int t0 = 0; t0 < m; t0 += 16) {
                if (t0) load_grp(base, m, t0, g_cur);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (t0 + 4 * u >= m) break;                             // (uniform)
                    const bool in = t0 + 4 * u + kk < m;
                    const double dx = g_cur.dxy[u].x, dy = g_cur.dxy[u].y, e = g_cur.et[u].x;
                    const uint32_t keyu = (uint32_t)__double2hiint(g_cur.et[u].y);      // (packed view: tail = {pixel, pair key})
                    double w = 1.0;
                    if (p.irls == 2) w = 1.0 / (1.0 + p.eta * e * e);
                    else if (p.irls == 1) { const double a = fabs(e); w = (a < p.eta) ? 1.0 : p.eta / a; }
                    const int bc = 3 * (int)(keyu >> 16), bp = 3 * (int)(keyu & 0xFFFFu);
                    if (in) {
                        rows_mask |= (1ull << ((bc >> 6) & 63)) | (1ull << (((bc + 5) >> 6) & 63)) | (1ull << ((bp >> 6) & 63)) | (1ull << (((bp + 5) >> 6) & 63));
                        rmin = min(rmin, min(bc, bp)); rmax = max(rmax, max(bc, bp) + 5);
                    }
                    const double wx = w * g_cur.x[u];                           // Yi_inv * dM_ddrot^T, model.cpp:483-487 / 679-683
                    if (in && el < 12) {
                        const int row = (el < 6) ? bc + el : bp + el - 6;
                        atomicAdd(&c0[row], wx * dx);
                        atomicAdd(&c1[row], wx * dy);
                    }
                }
            }
        }
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { rows_mask |= __shfl_xor(rows_mask, o); rmin = min(rmin, __shfl_xor(rmin, o)); rmax = max(rmax, __shfl_xor(rmax, o)); }
        // Only the 64-row blocks between the pixel's first and last touched row are written (a band over a long window: the pixel is in view
        // for a fraction of it); the SYRK masks everything else of these two columns out (SyrkParams::range), whatever the buffer holds, and
        // skips the 16 x 16 tiles in which no column of a quad has a row.
        const int lo16 = (rmax >= 0) ? rmin >> 4 : 1, hi16 = (rmax >= 0) ? rmax >> 4 : 0;
        const int lo = (rmax >= 0) ? lo16 >> 2 : 1, hi = (rmax >= 0) ? hi16 >> 2 : 0;
        const int r0 = 64 * lo, r1 = (64 * (hi + 1) < p.n) ? 64 * (hi + 1) : p.n;
        const double ic00 = rs0, ic11 = rs1;      // (the reciprocals of the factor's diagonal: the band loop had two divisions per row)
        // U = A12 * C^-T :  u0 = a0/c00 ;  u1 = (a1 - a0*c10/c00)/c11 ;  and the block's share of U y (the right-hand side b1 - U y)
        double* u0 = p.U + (size_t)p.ldu * (2 * (i - p.p0));
        double* u1 = u0 + p.ldu;
        for (int r = r0 + lane; r < r1; r += 64) {
            const double a0 = c0[r], a1 = c1[r];
            const double t0 = a0 * ic00, t1 = (a1 - t0 * c10) * ic11;
            u0[r] = t0;
            u1[r] = t1;
            c0[r] = 0.0; c1[r] = 0.0;
            const double uy = t0 * y0 + t1 * y1;
            if (uy != 0.0) atomicAdd(&s_rhs[r], uy);
        }
        if (lane == 0) {
            p.range[i - p.p0] = (uint16_t)(lo16 | (hi16 << 8));
            if (p.slice_mask && rows_mask) atomicOr(p.slice_mask + (i - p.p0) / kSyrkSlicePix, rows_mask);
        }
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        h_cur = h_nxt; h_nxt = h_nn; g_cur = g_nxt;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < p.n; r += 64 * kBuildWaves) {
        const double v = s_rhs[r];
        if (v != 0.0) atomicAdd(p.rhs_row + (size_t)p.lds * r, -v);
    }
}

// Column order of U (round 4, late).  The block-sparse SYRK forms a (row-block pair, slice) product wherever a slice of 128 consecutive columns' pixels has rows in
// both blocks, so what it costs is set by the UNION band of a slice.  In ascending panorama index (the reference's column order) a slice is a piece of a panorama
// row, 128+ pixels wide: under a panning camera its pixels are seen one after the other and the union band is 13 groups of 16 rows where a single pixel has 7.5
// (config 2's shape).  Ordered by panorama COLUMN first a slice is a piece of a column (or of a few neighbouring ones), whose pixels are seen together:
// 7.9 -> 5.8 products per slice there.
// Any order gives the same S; x2, y and the 2x2 factors stay indexed by the compact pixel.
// Built by counting, not sorting: every pixel takes a ticket in its panorama column (any order inside a column will do — its pixels have the same band),
// the W column counts are scanned, the pixel goes to offset[column] + ticket.
__global__ void emba_perm_ticket_kernel(const uint32_t* __restrict__ active_idx, long P, int W, uint32_t* __restrict__ col_cnt, uint32_t* __restrict__ ticket)
{
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    ticket[k] = atomicAdd(col_cnt + active_idx[k] % (uint32_t)W, 1u);
}
__global__ void emba_perm_place_kernel(const uint32_t* __restrict__ active_idx, long P, int W, const uint32_t* __restrict__ col_off, const uint32_t* __restrict__ ticket,
                                       uint32_t* __restrict__ perm)
{
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    perm[col_off[active_idx[k] % (uint32_t)W] + ticket[k]] = (uint32_t)k;
}

// Augmented system matrix (n+1) x (n+1), leading dimension lds: [A11 + lambda*diag(A11), . ; b1^T, 0] — the SYRK with the
// augmented U then leaves S in the leading n x n lower triangle and rhs = b1 - U y in row n.
__global__ void emba_schur_init_kernel(const double* __restrict__ A11, const double* __restrict__ b1, int n, double lambda,
                                       double* __restrict__ S, long lds)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)n * n) {
        const int r = (int)(i % n), c = (int)(i / n);
        const double a = A11[i];
        S[(size_t)lds * c + r] = (r == c) ? a + lambda * a : a;       // model.cpp:728-730
    }
    if (i < n) S[(size_t)lds * i + n] = b1[i];
    if (i == 0) S[(size_t)lds * n + n] = 0.0;
}

// rhs[c] = S_aug[n][c] (row n of the augmented matrix) for c >= skip, 0 before
__global__ void emba_schur_rhs_kernel(const double* __restrict__ S, long lds, int n, int skip, double* __restrict__ rhs)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) rhs[c] = (c >= skip) ? S[(size_t)lds * c + n] : 0.0;
}

// ---- split-K SYRK on the fp64 matrix cores ------------------------------------------------------------------------------
// C (lower triangle, column-major, ldc) -= A A^T for A = n x k column-major (lda).  Output is cut into 64x64 blocks (block
// pairs I >= J); grid.x = block pair, grid.y = K slice; the 4 waves of a block split the slice, every wave keeps its 4x4 tiles
// of 16x16 in registers (lane l holds A[I0+16ib+(l&15)][col 4s+(l>>4)] — the A/B operand layout — straight from memory),
// the block combines its waves in LDS and either subtracts from C directly (one slice) or writes a partial slab that
// emba_syrk_reduce_kernel sums (many slices: all the atomics of a direct update would land on the same n^2 words).
struct SyrkParams {
    const double* A; long lda; int n; long k; double* C; long ldc; double* slab; int nbp; int direct;
    // block-sparse form: blockIdx.y = part; the block walks the slices list[bp * n_slices + part], + gridDim.y, ... (count[bp] of them):
    // the column slices (2 * kSyrkSlicePix columns each) in which BOTH of its 64-row blocks have non-zeros
    const uint32_t* list; const uint32_t* count; int n_slices;
    const uint16_t* range;   // per PAIR of columns (one pixel): the 16-row groups [lo, hi] = (r & 255, r >> 8) that hold data; everything else of those columns reads as zero.  nullptr: all rows
    // ITEM form (round 5; items != nullptr, 1-D grid): one workgroup per (block pair, CHUNK of item_chunk consecutive slices) that holds at least one product, the
    // items in CHUNK-MAJOR order — the workgroups in flight at any time work on neighbouring slices, so the operand blocks the pairs of a slice share (each is read by
    // every pair of its band: B^2 block reads per slice for B blocks of data) are re-read while they are still in the Infinity Cache instead of from HBM at random
    // times of the launch.  Every item writes its 64 x 64 partial to slab[item]; emba_syrk_item_reduce_kernel sums a pair's items.
    // (Measured and dropped on the way: contiguous slice ranges per part with all pairs of a part on one XCD — with U's columns in panorama-column order a range of
    // slices is a time window, only the ~10 pairs of its band have work and the other 45 workgroups of the part idle: SYRK 1.22 -> 4.65 ms at config 2's shape.)
    const uint32_t* items; const uint32_t* n_items; const unsigned long long* slice_mask; int item_chunk; uint32_t item_cap;
};

// per block pair (I >= J) the slices whose columns touch both row blocks: one wave per pair, ballot-compacted
__global__ __launch_bounds__(64) void emba_syrk_lists_kernel(const unsigned long long* __restrict__ slice_mask, int n_slices, int nbp, uint32_t* __restrict__ list,
                                                             uint32_t* __restrict__ count)
{
    const int bp = blockIdx.x, lane = threadIdx.x;
    if (bp >= nbp) return;
    int I = (int)((sqrt(8.0 * bp + 1.0) - 1.0) * 0.5);
    while ((long)I * (I + 1) / 2 > bp) --I;
    while ((long)(I + 1) * (I + 2) / 2 <= bp) ++I;
    const int J = bp - I * (I + 1) / 2;
    const unsigned long long need = (1ull << (I & 63)) | (1ull << (J & 63));
    uint32_t n = 0;
    for (int s0 = 0; s0 < n_slices; s0 += 64) {
        const int sl = s0 + lane;
        const bool act = sl < n_slices && (slice_mask[sl] & need) == need;
        const unsigned long long m = __ballot(act);
        if (act) list[(size_t)bp * n_slices + n + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)sl;
        n += (uint32_t)__popcll(m);
    }
    if (lane == 0) count[bp] = n;
}

__device__ __forceinline__ void syrk_block_pair(int bp, int& I, int& J)
{   // bp -> (I, J), I >= J, row-major over the lower triangle
    int i = (int)((sqrt(8.0 * bp + 1.0) - 1.0) * 0.5);
    while ((long)i * (i + 1) / 2 > bp) --i;
    while ((long)(i + 1) * (i + 2) / 2 <= bp) ++i;
    I = i; J = bp - i * (i + 1) / 2;
}

__global__ __launch_bounds__(256) void emba_syrk_kernel(SyrkParams p)
{
    __shared__ double s_tile[64 * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int el = lane & 15, kk = lane >> 4;
    int I, J;
    __shared__ uint8_t s_vs[64];             // item form: the chunk's slices that hold a product of this pair, in order
    unsigned bx = blockIdx.x, by = blockIdx.y;
    uint32_t item_chunk_idx = 0; int item_ns = 0;
    if (p.items) {
        if (blockIdx.x >= p.n_items[0]) return;                               // (block-uniform; the grid is an upper bound)
        const uint32_t it = p.items[blockIdx.x];
        bx = it & 0xFFFFu; item_chunk_idx = it >> 16; by = 0;
    }
    syrk_block_pair((int)bx, I, J);
    if (p.items) {
        const unsigned long long need = (1ull << (I & 63)) | (1ull << (J & 63));
        const long sl = (long)item_chunk_idx * p.item_chunk + lane;
        const bool v = lane < p.item_chunk && sl < p.n_slices && (p.slice_mask[sl] & need) == need;
        const unsigned long long vm = __ballot(v);
        if (wv == 0 && v) s_vs[__popcll(vm & ((1ull << lane) - 1ull))] = (uint8_t)lane;
        item_ns = (int)__popcll(vm);
        __syncthreads();
    }
    const int I0 = 64 * I, J0 = 64 * J;
    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
    // The wave's column quads, as one flat sequence of nq: dense form — its share of the block's K slice; block-sparse form — its share of every
    // listed slice this block takes (blockIdx.y, + gridDim.y, ...), slice after slice.  quad(it) = first column of the wave's it-th quad
    // (all of it scalar: wave-uniform).
    constexpr int kQuadsPerSlice = 2 * kSyrkSlicePix / 16;
    const int wvs = __builtin_amdgcn_readfirstlane(wv);
    const uint32_t cnt = p.list ? p.count[bx] : 0u;
    const uint32_t* lst = p.list ? p.list + (size_t)bx * p.n_slices : nullptr;
    const long kslice = ((p.k + gridDim.y - 1) / gridDim.y + 3) / 4 * 4;
    const long dkb = (long)by * kslice, dke = (dkb + kslice < p.k) ? dkb + kslice : p.k;
    uint32_t l0 = by, lstep = gridDim.y;       // block-sparse: the block's entries of its pair's list are l0, l0 + lstep, ... (strided parts), or a contiguous piece of it
    int nq;
    if (p.items) nq = item_ns * kQuadsPerSlice;
    else if (p.list) nq = (cnt > by) ? (int)((cnt - by + gridDim.y - 1) / gridDim.y) * kQuadsPerSlice : 0;
    else { const long span = dke - dkb - 4 * wvs; nq = span > 0 ? (int)((span + 15) / 16) : 0; }
    auto quad = [&](int it) -> long {
        if (it >= nq) return -1;
        if (p.items) return ((long)item_chunk_idx * p.item_chunk + s_vs[it / kQuadsPerSlice]) * (2 * kSyrkSlicePix) + 16 * (it % kQuadsPerSlice) + 4 * wvs;
        if (p.list) return (long)lst[l0 + (uint32_t)(it / kQuadsPerSlice) * lstep] * (2 * kSyrkSlicePix) + 16 * (it % kQuadsPerSlice) + 4 * wvs;
        return dkb + 16L * it + 4 * wvs;
    };
    const long kend = (p.list || p.items) ? p.k : dke;
    // Operand loads are UNCONDITIONAL (clamped addresses, zeros selected afterwards) and run TWO quads ahead of the MFMAs through three
    // register buffers used round-robin: a wave's 16 MFMAs per quad are 1024 matrix-pipe cycles, two waves share a SIMD, so two quads in
    // flight cover ~2 us of load latency.  (Before: load, wait, 16 MFMAs, next load — 26 TFLOP/s dense, 9 block-sparse.)
    // (validity is ANDed into the operand's bits when it is USED, an iteration or two after the load: a select at the load lets the compiler
    // sink the load under the condition — a load under a lane mask is waited for at the end of its branch, vmcnt(0) after every second
    // load in the generated code — and a mask applied at the load is a use of the value, waited for on the spot)
    int rma[4], rmb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { rma[t] = (I0 + 16 * t + el < p.n) ? -1 : 0; rmb[t] = (J0 + 16 * t + el < p.n) ? -1 : 0; }
    // DIAG (I == J): the B operand IS the A operand — half the loads of such a pair (4 of the 10 pairs of a 4-block band) — and the tiles above
    // the diagonal are not formed
    auto run = [&](auto diag_c) {
        constexpr bool DIAG = decltype(diag_c)::value;
        auto load = [&](int it, double* av, double* bv, int& cm) {
            const long c4 = quad(it);
            const long col = c4 + kk;
            const bool cok = c4 >= 0 && col < kend;
            cm = cok ? -1 : 0;
            if (p.range) cm = cok ? (int)p.range[col >> 1] : 0x0001;   // (the raw range word rides with the stage; decoded in mma.  lo = 1 > hi = 0: nothing valid)
            const double* colp = p.A + (size_t)p.lda * (cok ? col : 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ri = I0 + 16 * t + el, rj = J0 + 16 * t + el;
                av[t] = colp[rma[t] ? ri : 0];
                if (!DIAG) bv[t] = colp[rmb[t] ? rj : 0];
            }
        };
        auto mma = [&](const double* av, const double* bv, int cm) {
            double am[4], bm[4];
            bool any_a[4], any_b[4];
            const int lo = cm & 255, hi = (cm >> 8) & 255;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int ma = cm & rma[t], mb = cm & rmb[t];
                if (p.range) {     // the lane's column has data in the 16-row group of this tile row?
                    ma = (4 * I + t >= lo && 4 * I + t <= hi) ? rma[t] : 0;
                    mb = (4 * J + t >= lo && 4 * J + t <= hi) ? rmb[t] : 0;
                }
                am[t] = __hiloint2double(__double2hiint(av[t]) & ma, __double2loint(av[t]) & ma);
                if (!DIAG) bm[t] = __hiloint2double(__double2hiint(bv[t]) & mb, __double2loint(bv[t]) & mb);
                any_a[t] = !p.range || __ballot(ma != 0) != 0ull;       // (wave-uniform) some column of the quad has rows in this 16-row group
                any_b[t] = DIAG ? any_a[t] : (!p.range || __ballot(mb != 0) != 0ull);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (DIAG && a < b) continue;
                    if (any_a[a] && any_b[b]) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[a], DIAG ? am[b] : bm[b], acc[a][b], 0, 0, 0);
                }
        };
        double a0[4], b0[4], a1[4], b1[4], a2[4], b2[4];
        int m0, m1, m2;
        load(0, a0, b0, m0); load(1, a1, b1, m1);
        // (one loop body of three stages and nothing else: with early exits between the stages the register allocator kept the 128 accumulator
        // registers in different places on different paths and copied them every trip; stages past the end run on zeros.  Round 4: a fourth
        // buffer, three quads ahead, changed nothing — 1.73 vs 1.69 ms at config 2's shape: the kernel is not short of bytes in flight; at the
        // MI355X's fp64 matrix rate, 64 cycles per 16x16x4, its 27 M MFMAs alone are 0.7 ms, its 6 GB of operand reads 0.75 ms.  Also measured: 16 B per lane
        // — a lane fetches the row pair (32u + 2el, + 1) and the halves feed two row-permuted MFMA tiles, half the load instructions, 256-B runs —
        // correct and no faster: 1.51 vs 1.50 ms, dense 1.23 vs 1.22)
        for (int it = 0; it < nq; it += 3) {
            load(it + 2, a2, b2, m2); mma(a0, b0, m0);
            load(it + 3, a0, b0, m0); mma(a1, b1, m1);
            load(it + 4, a1, b1, m1); mma(a2, b2, m2);
        }
    };
    if (I == J) run(std::true_type{}); else run(std::false_type{});
    for (int i = threadIdx.x; i < 64 * 64; i += 256) s_tile[i] = 0.0;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * a + kk + 4 * r, colt = 16 * b + el;      // C/D layout of v_mfma_f64_16x16x4_f64
                atomicAdd(&s_tile[colt * 64 + row], acc[a][b][r]);
            }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int row = I0 + (i & 63), colg = J0 + (i >> 6);
        if (p.items) {
            if (blockIdx.x < p.item_cap) p.slab[(size_t)blockIdx.x * 4096 + i] = s_tile[i];
            else if (row < p.n && colg < p.n && row >= colg && s_tile[i] != 0.0) atomicAdd(&p.C[(size_t)p.ldc * colg + row], -s_tile[i]);      // (more items than slabs: never at the sizes measured)
        }
        else if (p.direct) { if (row < p.n && colg < p.n && row >= colg) p.C[(size_t)p.ldc * colg + row] -= s_tile[i]; }
        else p.slab[((size_t)by * p.nbp + bx) * 4096 + i] = s_tile[i];
    }
}

// ---- the ITEM form's lists ----------------------------------------------------------------------------------------------------------------
// (1) flags: one workgroup per chunk, its threads over the block pairs — does chunk c hold a product of pair bp?  Then one thread per pair turns its flags
// into ordinals: ord[c * nbp + bp] = the pair's k-th item (k = 0, 1, ...), -1 elsewhere.  (As ONE kernel — a thread per pair walking chunks x slices — this
// took 418 us at config 2's shape.)
__global__ __launch_bounds__(256) void emba_syrk_item_flag_kernel(const unsigned long long* __restrict__ slice_mask, int n_slices, int nbp, int chunk, int32_t* __restrict__ ord)
{
    __shared__ unsigned long long s_m[64];
    const int c = blockIdx.x;
    const int s0 = c * chunk, ns = min(chunk, n_slices - s0);
    if ((int)threadIdx.x < ns) s_m[threadIdx.x] = slice_mask[s0 + threadIdx.x];
    __syncthreads();
    for (int bp = threadIdx.x; bp < nbp; bp += blockDim.x) {
        int I, J;
        syrk_block_pair(bp, I, J);
        const unsigned long long need = (1ull << (I & 63)) | (1ull << (J & 63));
        bool any = false;
        for (int k = 0; k < ns; ++k) any = any || ((s_m[k] & need) == need);
        ord[(size_t)c * nbp + bp] = any ? 1 : -1;
    }
}
__global__ void emba_syrk_item_ord_kernel(int nbp, int n_chunks, int32_t* __restrict__ ord, uint32_t* __restrict__ pair_cnt)
{
    const int bp = blockIdx.x * blockDim.x + threadIdx.x;
    if (bp >= nbp) return;
    int k = 0;
    for (int c = 0; c < n_chunks; ++c) {
        const size_t f = (size_t)c * nbp + bp;
        const bool any = ord[f] > 0;
        ord[f] = any ? k : -1;
        k += any ? 1 : 0;
    }
    pair_cnt[bp] = (uint32_t)k;
}
// (2) ordered compaction in chunk-major order (one workgroup): items[j] = pair | chunk << 16, pair_items[bp * n_chunks + k] = j, n_items
__global__ __launch_bounds__(1024) void emba_syrk_item_list_kernel(const int32_t* __restrict__ ord, int nbp, int n_chunks, uint32_t* __restrict__ items, uint32_t* __restrict__ pair_items,
                                                                   uint32_t* __restrict__ n_items)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_base;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) s_base = 0;
    __syncthreads();
    const long total = (long)nbp * n_chunks;
    for (long f0 = 0; f0 < total; f0 += 1024) {
        const long f = f0 + t;
        const int32_t k = f < total ? ord[f] : -1;
        const bool v = k >= 0;
        const unsigned long long m = __ballot(v);
        if (lane == 0) s_w[wv] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t off = s_base;
        for (int w = 0; w < wv; ++w) off += s_w[w];
        if (v) {
            const uint32_t j = off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            const uint32_t bp = (uint32_t)(f % nbp), c = (uint32_t)(f / nbp);
            items[j] = bp | (c << 16);
            pair_items[(size_t)bp * n_chunks + k] = j;
        }
        __syncthreads();
        if (t == 0) { uint32_t tot = 0; for (int w = 0; w < 16; ++w) tot += s_w[w]; s_base += tot; }
        __syncthreads();
    }
    if (t == 0) n_items[0] = s_base;
}
// (3) C -= the sum of a pair's item slabs
__global__ void emba_syrk_item_reduce_kernel(const double* __restrict__ slab, const uint32_t* __restrict__ pair_items, const uint32_t* __restrict__ pair_cnt, int n_chunks,
                                             uint32_t item_cap, int nbp, int n, double* __restrict__ C, long ldc)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)nbp * 4096) return;
    const int bp = (int)(idx >> 12), i = (int)(idx & 4095);
    int I, J;
    syrk_block_pair(bp, I, J);
    const int row = 64 * I + (i & 63), col = 64 * J + (i >> 6);
    if (row >= n || col >= n || row < col) return;
    const uint32_t cnt = pair_cnt[bp];
    const uint32_t* lst = pair_items + (size_t)bp * n_chunks;
    double a0 = 0.0, a1 = 0.0;
    uint32_t k = 0;
    for (; k + 1 < cnt; k += 2) {
        const uint32_t j0 = lst[k], j1 = lst[k + 1];
        if (j0 < item_cap) a0 += slab[(size_t)j0 * 4096 + i];
        if (j1 < item_cap) a1 += slab[(size_t)j1 * 4096 + i];
    }
    if (k < cnt) { const uint32_t j0 = lst[k]; if (j0 < item_cap) a0 += slab[(size_t)j0 * 4096 + i]; }
    C[(size_t)ldc * col + row] -= a0 + a1;
}

// C -= sum over the split-K slabs.  blockIdx.y takes a group of kSyrkReduceGroup slabs, so the reduction of a small matrix
// (K = 21: ONE 64 x 64 tile but 1024 slabs) is spread over hundreds of blocks instead of 16; groups meet through fp64 atomics
// (4096 x groups adds: far below the atomic rate).
constexpr int kSyrkReduceGroup = 16;
__global__ void emba_syrk_reduce_kernel(const double* __restrict__ slab, int nks, int nbp, int n, double* __restrict__ C, long ldc)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)nbp * 4096) return;
    const int bp = (int)(idx >> 12), i = (int)(idx & 4095);
    int I, J;
    syrk_block_pair(bp, I, J);
    const int row = 64 * I + (i & 63), col = 64 * J + (i >> 6);
    if (row >= n || col >= n || row < col) return;
    const int k0 = blockIdx.y * kSyrkReduceGroup, k1 = (k0 + kSyrkReduceGroup < nks) ? k0 + kSyrkReduceGroup : nks;
    double acc = 0.0;
    for (int ks = k0; ks < k1; ++ks) acc += slab[((size_t)ks * nbp + bp) * 4096 + i];
    if (gridDim.y == 1) C[(size_t)ldc * col + row] -= acc;
    else atomicAdd(&C[(size_t)ldc * col + row], -acc);
}

// ---- blocked Cholesky (lower, column-major), panels of 64 ---------------------------------------------------------------
// (1) factor the diagonal block.
// The factorisation proper, by FOUR waves: every wave holds all 64 rows (lane r = row r) of 16 of the block's columns — wave w the columns 16w .. 16w+15 —,
// the owner of column j scales it (pivot as rsqrt_nr) and puts it in LDS, one barrier, then every wave with columns right of j updates them from broadcast
// reads.  Right-looking, every update pinned where it is written (an empty asm with the register as in/out operand: left alone the compiler sinks the updates
// to the step that reads the column — the left-looking form — and keeps all the broadcast values alive until then), compile-time loops (every register index
// static).  History: one wave, left-looking on v_readlane pairs, sqrt + division per column: 35 us per panel; one wave, right-looking through LDS: 28 (it
// issued ~15 k instructions); four waves issue a quarter each and the 64 barriers cost less than that: 18.  rw[cc] = the lane's row, column
// 16w + cc; on return rw[cc] = L[r][16w + cc] for 16w + cc <= r < nb.  All 256 threads of the block must call it.
__device__ __forceinline__ void chol_diag_block4(double (&rw)[16], double dg0, double (*s_col)[64], int r, int w, int nb, int* __restrict__ info)
{
    bool bad = false, bad_sign = false;
    // column j from the lanes' current row[j]: the pivot's square root and the scaled column
    auto column = [&](int j, double rj) -> double {
        const double d = readlane_f64(rj, j);
        // A vanishing pivot does not stop the reference: Eigen's ldlt (model.cpp:789) leaves such a column as it is and its solve takes
        // the PSEUDO-inverse of D — a zero update in that component (LDLT.h:362-381, 583-589; pinned: tests/golden/eigen_solvers.npz).
        // The case that occurs in practice is a control pose no event constrains: its rows and columns of S are exactly zero, so d == 0
        // here whatever the pivot order.  Same behaviour: the column is zeroed, L[j][j] = 0 marks it, the substitutions return 0 there.
        // (round 4) Only a VANISHING pivot is that case: d == 0, or a non-positive value within rounding of it — 64 ulp of the
        // diagonal entry it was eliminated from.  A pivot that is clearly negative, or not finite, means S is indefinite or carries a NaN: Eigen
        // factors the former with a negative D entry and propagates the latter; a Cholesky factorisation can do neither, and a finite, partly
        // zeroed x1 with EMBA_OK would hide corrupted equations — info bit 2 (4), which the solve returns as EMBA_ERR_NUMERIC (the LM loop then
        // rejects the step, as it does for the NaN cost the reference would see).
        const double d0 = readlane_f64(dg0, j);
        // (a positive SUBNORMAL pivot is not usable either — rsqrt_nr's y * y overflows on it and the column would turn into inf / NaN
        // with no info bit set; Eigen's own tolerance for "D_j is zero" is DBL_MIN, LDLT.h:583-589 — it is a vanishing pivot)
        const bool ok = d >= 2.2250738585072014e-308 && d < 1.7e308;                       // (inf / NaN: not ok, not vanishing)
        const bool vanishing = !ok && (d < 2.2250738585072014e-308) && (d >= -64.0 * 2.220446049250313e-16 * fabs(d0));      // false for NaN
        bad |= (j < nb) && !ok;
        bad_sign |= (j < nb) && !ok && !vanishing;
        const double rs = ok ? rsqrt_nr(d) : 0.0;
        double piv = d * rs;
        piv = ok ? fma(0.5 * rs, fma(-piv, piv, d), piv) : 0.0;                             // sqrt(d), one correction step
        return (r == j) ? piv : rj * rs;                     // rows above the diagonal hold garbage that is never read (only entries c > j are)
    };
    static_for<0, 64>([&](auto jc) {
        constexpr int j = decltype(jc)::value, owner = j >> 4, jj = j & 15;
        double* col = s_col[j & 1];
        double lj = 0.0;
        if (w == owner) { lj = column(j, rw[jj]); rw[jj] = lj; col[r] = lj; }
        __syncthreads();
        if (w == owner) {
            if constexpr (jj < 15) {
                double2 lc[8];
                static_for<0, 8>([&](auto qc) { constexpr int q = decltype(qc)::value; if constexpr (2 * q + 1 > jj) lc[q] = reinterpret_cast<const double2*>(col + 16 * owner)[q]; });
                static_for<0, 8>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    if constexpr (2 * q > jj) { rw[2 * q] -= lj * lc[q].x; __asm__ volatile("" : "+v"(rw[2 * q])); }
                    if constexpr (2 * q + 1 > jj) { rw[2 * q + 1] -= lj * lc[q].y; __asm__ volatile("" : "+v"(rw[2 * q + 1])); }
                });
            }
        } else if (w > owner) {
            const double lr = col[r];
            double2 lc[8];
            static_for<0, 8>([&](auto qc) { constexpr int q = decltype(qc)::value; lc[q] = reinterpret_cast<const double2*>(col + 16 * w)[q]; });
            static_for<0, 8>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                rw[2 * q] -= lr * lc[q].x; __asm__ volatile("" : "+v"(rw[2 * q]));
                rw[2 * q + 1] -= lr * lc[q].y; __asm__ volatile("" : "+v"(rw[2 * q + 1]));
            });
        }
    });
    if (bad && r == 0) atomicOr(info, 2);     // diagnostic only (what ldlt.info() == NumericalIssue is to the reference: never read)
    if (bad_sign && r == 0) atomicOr(info, 4);
}

__global__ __launch_bounds__(256) void emba_chol_diag_kernel(double* __restrict__ A, long ld, int jb, int nb, int* __restrict__ info)
{
    __shared__ __attribute__((aligned(16))) double s_col[2][64];
    const int r = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double rw[16];
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) { const int c = 16 * w + cc; rw[cc] = (r < nb && c < nb) ? A[(size_t)ld * (jb + c) + jb + r] : ((r == c) ? 1.0 : 0.0); }   // identity padding
    const double dg0 = (r < nb) ? A[(size_t)ld * (jb + r) + jb + r] : 1.0;      // the diagonal entry before elimination (the scale of the pivot test)
    chol_diag_block4(rw, dg0, s_col, r, w, nb, info);
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) { const int c = 16 * w + cc; if (c < nb && r >= c && r < nb) A[(size_t)ld * (jb + c) + jb + r] = rw[cc]; }
}

// (2) panel below the diagonal block: row * L_diag^-T for the rows A[jb+nb .. n) of the panel.  Lanes are the panel's 64 COLUMNS; lane c keeps
// L_diag[c][0..c) in registers; a wave carries kTrsmRows rows at a time: x_k = a_k / L[k][k] leaves lane k by v_readlane and updates the lanes
// c > k.  (Until round 3: one thread per row, 2016 dependent fmas on LDS broadcast reads — 35 us per panel whatever the number of rows.)
constexpr int kTrsmRows = 4;
__global__ __launch_bounds__(256) void emba_chol_trsm_kernel(double* __restrict__ A, long ld, int n, int jb, int nb)
{
    const int c = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long r0 = (long)jb + nb + wave * kTrsmRows;
    if (r0 >= n) return;                                                     // (wave-uniform)
    double Lm[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) Lm[k] = (c < nb && k < c) ? A[(size_t)ld * (jb + k) + jb + c] : 0.0;      // L_diag[c][k], strictly below the diagonal
    const double dg = (c < nb) ? A[(size_t)ld * (jb + c) + jb + c] : 0.0;
    const double rd = (dg != 0.0) ? 1.0 / dg : 0.0;                           // zeroed column of a vanishing pivot (emba_chol_diag_kernel): x = 0
    double a[kTrsmRows];
#pragma unroll
    for (int i = 0; i < kTrsmRows; ++i) a[i] = (c < nb && r0 + i < n) ? A[(size_t)ld * (jb + c) + r0 + i] : 0.0;
#pragma unroll
    for (int k = 0; k < 64; ++k) {
#pragma unroll
        for (int i = 0; i < kTrsmRows; ++i) {
            const double xk = readlane_f64(a[i] * rd, k);
            a[i] = fma(-xk, Lm[k], a[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < kTrsmRows; ++i) if (c < nb && r0 + i < n) A[(size_t)ld * (jb + c) + r0 + i] = a[i] * rd;
}

// Small systems, m + 1 <= 64 (K <= 21 with the first pose fixed): factor, forward and backward substitution in ONE wave and one launch.  The
// right-hand side is row m of the augmented block (schur_factor_solve): factored along as one more row it becomes z = L^-1 rhs; L goes to LDS
// and L^T x = z is solved as in emba_chol_trsv_kernel.  x (m entries) -> x_out[skip ..), zeros in front.  (Until round 4: diagonal factor, panel
// solve of the one rhs row, rhs copy, triangular solve — four launches, 58 us on the device at K = 21.)
__global__ __launch_bounds__(256) void emba_chol_small_kernel(const double* __restrict__ A, long ld, int m, int skip, double* __restrict__ x_out, int* __restrict__ info)
{
    __shared__ __attribute__((aligned(16))) double s_col[2][64];
    __shared__ double s_l[64 * 65];
    __shared__ double s_z[64];
    const int r = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double rw[16];
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) { const int c = 16 * w + cc; rw[cc] = (r <= m && c < m && r >= c) ? A[(size_t)ld * c + r] : ((r == c) ? 1.0 : 0.0); }   // rows 0..m (m: the rhs), columns 0..m-1; identity elsewhere
    const double dg0 = (r < m) ? A[(size_t)ld * r + r] : 1.0;
    chol_diag_block4(rw, dg0, s_col, r, w, m, info);
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
        const int c = 16 * w + cc;
        s_l[c * 65 + r] = (r < m && c <= r) ? rw[cc] : 0.0;                       // L[r][c]
        if (r == m) s_z[c] = (c < m) ? rw[cc] : 0.0;                              // z
    }
    __syncthreads();
    if (w != 0) return;
    double v = (r < m) ? s_z[r] : 0.0;
    const double dgt = s_l[r * 65 + r];
    const double rdt = (r < m && dgt != 0.0) ? 1.0 / dgt : 0.0;                   // pseudo-inverse: zero update where the pivot vanished
#pragma unroll
    for (int j = 63; j >= 0; --j) {
        const double xj = readlane_f64(v * rdt, j);
        const double ltj = (j > r) ? s_l[r * 65 + j] : 0.0;                       // L[j][r] = L^T[r][j]
        v = fma(-xj, ltj, v);
    }
    if (r < m) x_out[skip + r] = v * rdt;
    if (r < skip) x_out[r] = 0.0;
}

// (3) trailing update after a panel, and the NEXT panel's diagonal factor in the same launch.  X = the panel's rows below the diagonal block
// (after the panel solve; n_t of them, the right-hand-side row included), nb columns; the trailing matrix T (n_t x n_t, lower triangle) takes
// T -= X X^T in 64 x 64 tiles, one workgroup per tile (I >= J): both 64 x nb slabs of X staged in LDS, wave w forms rows 16w..16w+15 of the tile
// over the whole panel width (4 x 16 MFMAs).  The workgroup of tile (0, 0) then factors the next diagonal block from its updated tile (through
// LDS: its four waves, chol_diag_block4), which until round 4 was a launch of its own between this one and the next panel solve (28 + 27 us per panel in
// sequence; together 34).  nb_next = size of the next diagonal block (0: none — the tile (0, 0) then only holds the right-hand side row).
__global__ __launch_bounds__(256) void emba_chol_trail_kernel(double* __restrict__ A, long ld, int jb, int nb, int n_t, int nb_next, int* __restrict__ info)
{
    __shared__ __attribute__((aligned(16))) double s_i[64 * 64];
    __shared__ __attribute__((aligned(16))) double s_j[64 * 65];           // (tile (0, 0): I == J, this slab is free and carries the updated tile to the factor)
    __shared__ __attribute__((aligned(16))) double s_col[2][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, el = lane & 15, kk = lane >> 4;
    int I, J;
    syrk_block_pair(blockIdx.x, I, J);
    const int I0 = 64 * I, J0 = 64 * J;
    const double* X = A + (size_t)ld * jb + (jb + nb);                       // X[row][k] = X[ld * k + row]
    double* T = A + (size_t)ld * (jb + nb) + (jb + nb);
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int row = i & 63, k = i >> 6;
        s_i[i] = (k < nb && I0 + row < n_t) ? X[(size_t)ld * k + I0 + row] : 0.0;
        if (I != J) s_j[i] = (k < nb && J0 + row < n_t) ? X[(size_t)ld * k + J0 + row] : 0.0;
    }
    __syncthreads();
    const double* sb = (I != J) ? s_j : s_i;
    double4_t acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
        const int k = 4 * q + kk;
        const double av = s_i[k * 64 + 16 * wv + el];
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, sb[k * 64 + 16 * b + el], acc[b], 0, 0, 0);
    }
    const bool first = (I == 0 && J == 0 && nb_next > 0);
    if (first) __syncthreads();                                              // (s_j is about to be written: every wave is done reading s_i — the same slab here — no hazard, but keep the phases apart)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int row = 16 * wv + kk + 4 * r4, col = 16 * b + el;        // C/D layout of v_mfma_f64_16x16x4_f64
            const int gr = I0 + row, gc = J0 + col;
            if (gr < n_t && gc < n_t && gr >= gc) {
                double* tp = T + (size_t)ld * gc + gr;
                const double v = *tp - acc[b][r4];
                if (first && gr < nb_next) s_j[col * 65 + row] = v;          // the next diagonal block: to the factor below, which writes it
                else *tp = v;
            }
        }
    if (!first) return;
    __syncthreads();
    {
        const int r = lane, w = __builtin_amdgcn_readfirstlane(wv);
        double rw[16];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) { const int c = 16 * w + cc; rw[cc] = (r < nb_next && c < nb_next && r >= c) ? s_j[c * 65 + r] : ((r == c) ? 1.0 : 0.0); }   // (entries above the diagonal are never read)
        const double dg0 = (r < nb_next) ? s_j[r * 65 + r] : 1.0;
        chol_diag_block4(rw, dg0, s_col, r, w, nb_next, info);
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) { const int c = 16 * w + cc; if (c < nb_next && r >= c && r < nb_next) T[(size_t)ld * c + r] = rw[cc]; }
    }
}

// Solve L^T x = z in place (one workgroup; b holds z = L^-1 rhs: the factorisation carried the right-hand side along as an extra row,
// schur_factor_solve).  64-wide blocks from the bottom: one wave substitutes inside the block — lane t keeps column t of the block in registers,
// x_j leaves lane j by v_readlane —, then every thread takes one row above the block and subtracts its 64-term dot product (the row's 64 factor
// entries are contiguous: column r of L).
constexpr int kTrsvThreads = 512;  // (round 5: 1024 threads capped the kernel at 128 VGPRs and the single wave's 64 preloaded factor entries spilled: 120 B of scratch per lane)
constexpr int kTrsvMaxN = 3072;   // right-hand sides up to this length stay in LDS during the sweep (K <= 1024)
__global__ __launch_bounds__(kTrsvThreads) void emba_chol_trsv_kernel(const double* __restrict__ L, long ld, int n, double* __restrict__ b)
{
    __shared__ double s_l[64 * 65];
    __shared__ double s_x[64];
    __shared__ double s_b[kTrsvMaxN];
    const int t = threadIdx.x;
    const bool in_lds = n <= kTrsvMaxN;
    if (in_lds) { for (int i = t; i < n; i += kTrsvThreads) s_b[i] = b[i]; }
    __syncthreads();
    for (int jb = ((n - 1) / 64) * 64; jb >= 0; jb -= 64) {
        const int nb = (n - jb < 64) ? n - jb : 64;
        for (int i = t; i < 64 * 64; i += kTrsvThreads) { const int r = i & 63, c = i >> 6; s_l[c * 65 + r] = (r < nb && c < nb && r >= c) ? L[(size_t)ld * (jb + c) + jb + r] : 0.0; }
        __syncthreads();
        if (t < 64) {
            double v = (t < nb) ? (in_lds ? s_b[jb + t] : b[jb + t]) : 0.0;
            const double dgt = s_l[t * 65 + t];
            const double rdt = (dgt != 0.0) ? 1.0 / dgt : 0.0;                          // pseudo-inverse: zero update where the pivot vanished
#pragma unroll
            for (int j = 63; j >= 0; --j) {
                const double xj = readlane_f64(v * rdt, j);
                const double ltj = (j > t) ? s_l[t * 65 + j] : 0.0;                      // L[j][t] = L^T[t][j], strictly above the diagonal of L^T
                v = fma(-xj, ltj, v);
            }
            v *= rdt;
            s_x[t] = v;
            if (t < nb) { if (in_lds) s_b[jb + t] = v; else b[jb + t] = v; }
        }
        __syncthreads();
        for (int r = t; r < jb; r += kTrsvThreads) {
            const double* lp = L + (size_t)ld * r + jb;
            double d0 = 0.0, d1 = 0.0;
            if (nb == 64 && ((reinterpret_cast<uintptr_t>(lp) & 15) == 0)) {      // the row's 64 factor entries as 32 16-B loads
                const double2* lp2 = reinterpret_cast<const double2*>(lp);
#pragma unroll
                for (int c = 0; c < 32; ++c) { const double2 l2 = lp2[c]; d0 = fma(l2.x, s_x[2 * c], d0); d1 = fma(l2.y, s_x[2 * c + 1], d1); }
            } else if (nb == 64) {
#pragma unroll
                for (int c = 0; c < 64; c += 2) { d0 = fma(lp[c], s_x[c], d0); d1 = fma(lp[c + 1], s_x[c + 1], d1); }
            } else {
                for (int c = 0; c < nb; ++c) d0 = fma(lp[c], s_x[c], d0);
            }
            if (in_lds) s_b[r] -= d0 + d1; else b[r] -= d0 + d1;
        }
        __syncthreads();
    }
    if (in_lds) { for (int i = t; i < n; i += kTrsvThreads) b[i] = s_b[i]; }
}

// x2_i = C_i^-T (y_i - z_i), z_i = A12_i^T x1 C^-T... computed from the records of pixel i:  A12_i^T x1 = sum_m w_m (v_m . x1) dp_m,
// then z = C^-1 (A12_i^T x1)  (since U^T x1 = C^-1 A12^T x1)                                                       model.cpp:791
__global__ __launch_bounds__(256) void emba_schur_x2_kernel(RecView view, const uint32_t* __restrict__ off,
                                                            const double* __restrict__ yv, const double* __restrict__ cfac,
                                                            const double* __restrict__ x1, int irls, double eta, long P, double* __restrict__ x2, int n)
{
    // (round 5) x1 — 3K doubles — sits in LDS: a record's 12 products needed x1[row] by a gather that could only be issued once the record's pair key had
    // arrived, a second dependent round trip per group of records
    extern __shared__ double s_x1[];
    for (int r = threadIdx.x; r < n; r += 256) s_x1[r] = x1[r];
    __syncthreads();
    // (Round 4: prefetching the next pixel's first record group, as the U build does, made this kernel SLOWER — 374 -> 503 us at config 2's shape: at 56
    // VGPRs it runs eight waves per SIMD, which already overlap the pixels' round trips; the extra stage costs occupancy.)
    // One wave per pixel; the pixel's records are consecutive (emba_csr_fill_sorted_kernel), 16 of them in flight (lane l: element l&15 of record
    // l>>4 of each group of four).  The 12-term dot product of a record and its weight stay inside the record's DPP row of 16 lanes — row shifts
    // on the VALU, no LDS crossbar: with xor-shuffles and broadcasts (14 ds_bpermute per group) the kernel ran at the LDS pipe's rate.
    const int lane = threadIdx.x & 63, el = lane & 15, kk = lane >> 4;
    const long nwaves = (long)gridDim.x * 4;
    const long i_first = (long)blockIdx.x * 4 + (threadIdx.x >> 6), i_last = P - 1;
    auto load_b = [&](long i, uint32_t& b0, uint32_t& b1) { const long ic = i < i_last ? i : i_last; b0 = off[ic]; b1 = off[ic + 1]; };
    uint32_t cb0, cb1, nb0, nb1;
    load_b(i_first, cb0, cb1);
    for (long i = i_first; i < P; i += nwaves) {
        load_b(i + nwaves, nb0, nb1);                                  // (the next pixel's list bounds: in flight during this one)
        double acc = 0.0;                                              // lane 12 of a row: sum of w (v . x1) dx over its records; lane 13: ... dy
        const uint32_t b0 = cb0, b1 = cb1;
        for (uint32_t base = b0; base < b1; base += 16) {
            double xv[4], ev[4]; uint32_t key[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t r = base + 4 * u + kk;
                const uint32_t s = r < b1 ? r : b0;
                xv[u] = view.rec[(size_t)kRecStride * s + el];
                ev[u] = irls ? view.rec[(size_t)kRecStride * s + 14] : 0.0;     // the residual, for the IRLS weight (same line: one cached request per record)
                key[u] = rec_key(view, s);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (base + 4 * u >= b1) break;                             // (uniform)
                const bool in = base + 4 * u + kk < b1;
                const double x = xv[u];
                const int row = (el < 6) ? 3 * (int)(key[u] >> 16) + el : 3 * (int)(key[u] & 0xFFFFu) + el - 6;
                double d = (in && el < 12) ? x * s_x1[row] : 0.0;
                // rotations inside the row: every lane ends with the whole dot product.  (NOT a shift-scan with the total handed to lanes 12 / 13 by
                // two different shifts behind a select: the compiler turns that select into two branches with one DPP move in each, and a DPP
                // move reads zero from lanes its branch has masked off)
                d += dpp_row_ror<1>(d); d += dpp_row_ror<2>(d); d += dpp_row_ror<4>(d); d += dpp_row_ror<8>(d);
                const double dot = d, e = ev[u];
                double w = 1.0;
                if (irls == 2) w = 1.0 / (1.0 + eta * e * e);
                else if (irls == 1) { const double a = fabs(e); w = (a < eta) ? 1.0 : eta / a; }
                if (in && (el == 12 || el == 13)) acc += w * dot * x;      // x = dp_x in lane 12, dp_y in lane 13
            }
        }
        acc += __shfl_xor(acc, 16); acc += __shfl_xor(acc, 32);
        const double a0 = readlane_f64(acc, 12), a1 = readlane_f64(acc, 13);
        if (lane == 0) {
            const double c00 = cfac[3 * i], c10 = cfac[3 * i + 1], c11 = cfac[3 * i + 2];
            const double z0 = a0 / c00, z1 = (a1 - c10 * z0) / c11;          // z = C^-1 (A12_i^T x1)
            const double t0 = yv[2 * i] - z0, t1 = yv[2 * i + 1] - z1;
            const double bq = t1 / c11;                                      // x2 = C^-T t
            x2[2 * i + 1] = bq;
            x2[2 * i] = (t0 - c10 * bq) / c00;
        }
        cb0 = nb0; cb1 = nb1;
    }
}


// ---- LEGM::solveNormalEqCG (model.cpp:794-840): Eigen::ConjugateGradient on [A11m A12; A12^T A22m], matrix-free ---------------
// Vectors have n + 2P entries: [pose part (3K, entries of a fixed first pose stay 0) | map part (2 per active pixel)].  One
// application of the matrix reads every record once (through the per-pixel lists): A12^T v1 per pixel by dot products, A12 v2
// scattered into a per-block LDS copy of the pose part (<= 6 KB) that is flushed with fp64 atomics.
__global__ __launch_bounds__(256) void emba_cg_a11_kernel(const double* __restrict__ A11, int n, double lambda, int skip, const double* __restrict__ v,
                                                          double* __restrict__ y)
{   // one wave per row r: y[r] = sum_c A11m[r][c] v[c]
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    double acc = 0.0;
    if (r >= skip)
        for (int c = skip + lane; c < n; c += 64) { const double a = A11[(size_t)n * c + r]; acc += ((r == c) ? a + lambda * a : a) * v[c]; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) y[r] = acc;      // (= : this kernel runs first; the pixel kernel adds A12 v2 on top)
}

struct CgPixParams {
    RecView view; const uint32_t* off; const double* A22b2; double lambda; int irls; double eta; int n, skip; long P;
    const double* v; double* y;
};

// (round 6) In the x2 kernel's form: sixteen records of a pixel in flight per wave (lane l: element l & 15 of record l >> 4 of each group of four), the pose part of
// the vector in LDS, the 12-term dot product of a record summed by DPP row rotations, dp / e / the pair key of a record straight from its line (one cached request per
// 16-lane row).  Until then: four records per trip, v1[row] gathered from global memory behind the key, five ds_bpermute per group — 0.80 ms per application at
// config 2's shape (1.2 TB/s over the 0.96 GB of records), 80 % of every CG iteration.
__global__ __launch_bounds__(256) void emba_cg_pixel_kernel(CgPixParams p)
{
    extern __shared__ double s_cg[];     // [0, n): pose part contributed by this block; [n, 2n): the pose part of v
    double* s_y = s_cg; double* s_v1 = s_cg + p.n;
    for (int r = threadIdx.x; r < p.n; r += 256) { s_y[r] = 0.0; s_v1[r] = p.v[r]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, el = lane & 15, kk = lane >> 4;
    const long nwaves = (long)gridDim.x * 4;
    const double* v2 = p.v + p.n;
    const long i_first = (long)blockIdx.x * 4 + (threadIdx.x >> 6), i_last = p.P - 1;
    auto load_b = [&](long i, uint32_t& b0, uint32_t& b1) { const long ic = i < i_last ? i : i_last; b0 = p.off[ic]; b1 = p.off[ic + 1]; };
    uint32_t cb0, cb1, nb0, nb1;
    load_b(i_first, cb0, cb1);
    for (long i = i_first; i < p.P; i += nwaves) {
        load_b(i + nwaves, nb0, nb1);                                  // (the next pixel's list bounds: in flight during this one)
        const double vx = v2[2 * i], vy = v2[2 * i + 1];
        const double* q = p.A22b2 + 5 * i;
        const double q0 = q[0], q1 = q[1], q2 = q[2];                  // (requested in front of the records, answered with them)
        double a0 = 0.0, a1 = 0.0;
        const uint32_t b0 = cb0, b1 = cb1;
        for (uint32_t base = b0; base < b1; base += 16) {
            double xv[4]; double2 dxy[4], et[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t r = base + 4 * u + kk;
                const double* rp = p.view.rec + (size_t)kRecStride * (r < b1 ? r : b0);
                xv[u] = rp[el];
                dxy[u] = reinterpret_cast<const double2*>(rp)[6];
                et[u] = reinterpret_cast<const double2*>(rp)[7];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (base + 4 * u >= b1) break;                             // (uniform)
                const bool in = base + 4 * u + kk < b1;
                const double x = xv[u], dx = dxy[u].x, dy = dxy[u].y, e = et[u].x;
                const uint32_t key = (uint32_t)__double2hiint(et[u].y);   // (the records are in pixel order: tail = {pixel, pair key})
                const int row = (el < 6) ? 3 * (int)(key >> 16) + el : 3 * (int)(key & 0xFFFFu) + el - 6;
                const bool live = in && el < 12 && row >= p.skip;
                double d = live ? x * s_v1[row] : 0.0;
                d += dpp_row_ror<1>(d); d += dpp_row_ror<2>(d); d += dpp_row_ror<4>(d); d += dpp_row_ror<8>(d);      // every lane of the row: the record's dot product
                double w = 1.0;
                if (p.irls == 2) w = 1.0 / (1.0 + p.eta * e * e);
                else if (p.irls == 1) { const double a = fabs(e); w = (a < p.eta) ? 1.0 : p.eta / a; }
                if (in) { a0 += w * d * dx; a1 += w * d * dy; }            // identical in the 16 lanes of a record
                if (live) atomicAdd(&s_y[row], w * (dx * vx + dy * vy) * x);     // A12 v2
            }
        }
        a0 += __shfl_xor(a0, 16); a0 += __shfl_xor(a0, 32);
        a1 += __shfl_xor(a1, 16); a1 += __shfl_xor(a1, 32);
        if (lane == 0) {
            p.y[p.n + 2 * i] = a0 + (q0 + p.lambda * q0) * vx + q1 * vy;
            p.y[p.n + 2 * i + 1] = a1 + q1 * vx + (q2 + p.lambda * q2) * vy;
        }
        cb0 = nb0; cb1 = nb1;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < p.n; r += 256) { const double a = s_y[r]; if (a != 0.0) atomicAdd(p.y + r, a); }
}

// setup: r = b (first-pose rows zeroed), invd = 1/diag (DiagonalPreconditioner: 1 where the diagonal vanishes), x = 0, p = invd.*r;
// out[0] += r.r, out[1] += r.p
__global__ __launch_bounds__(256) void emba_cg_init_kernel(const double* __restrict__ A11, const double* __restrict__ b1, const double* __restrict__ A22b2,
                                                           int n, int skip, long P, double lambda, double* __restrict__ x, double* __restrict__ r,
                                                           double* __restrict__ pv, double* __restrict__ invd, double* __restrict__ out, long sum_from = 0)
{   // sum_from: first entry that counts in the two sums (the sharded solve: the pose part is replicated on every rank and must enter the all-reduced sums once)
    __shared__ double s_w[2][4];
    const long N = n + 2 * P;
    double rr = 0.0, rp = 0.0;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < N; k += (long)gridDim.x * 256) {
        double d, b;
        if (k < n) { const double a = A11[(size_t)n * k + k]; d = a + lambda * a; b = (k >= skip) ? b1[k] : 0.0; }
        else { const long i = (k - n) >> 1; const int h = (int)((k - n) & 1); const double* q = A22b2 + 5 * i; const double a = h ? q[2] : q[0]; d = a + lambda * a; b = q[3 + h]; }
        const double iv = (d != 0.0) ? 1.0 / d : 1.0;
        x[k] = 0.0; r[k] = b; invd[k] = iv; pv[k] = iv * b;
        if (k >= sum_from) { rr += b * b; rp += b * iv * b; }
    }
    rr = wave_sum(rr); rp = wave_sum(rp);
    if ((threadIdx.x & 63) == 0) { s_w[0][threadIdx.x >> 6] = rr; s_w[1][threadIdx.x >> 6] = rp; }
    __syncthreads();
    if (threadIdx.x == 0) { atomicAdd(out, (s_w[0][0] + s_w[0][1]) + (s_w[0][2] + s_w[0][3])); atomicAdd(out + 1, (s_w[1][0] + s_w[1][1]) + (s_w[1][2] + s_w[1][3])); }
}

__global__ __launch_bounds__(256) void emba_cg_dot_kernel(const double* __restrict__ a, const double* __restrict__ b, long N, double* __restrict__ out)
{
    __shared__ double s_w[4];
    double acc = 0.0;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < N; k += (long)gridDim.x * 256) acc += a[k] * b[k];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
}

// x += alpha p; r -= alpha t; out += r.r
__global__ __launch_bounds__(256) void emba_cg_xr_kernel(double alpha, const double* __restrict__ pv, const double* __restrict__ t, long N,
                                                         double* __restrict__ x, double* __restrict__ r, double* __restrict__ out, long sum_from = 0)
{
    __shared__ double s_w[4];
    double acc = 0.0;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < N; k += (long)gridDim.x * 256) {
        x[k] += alpha * pv[k];
        const double rv = r[k] - alpha * t[k];
        r[k] = rv; if (k >= sum_from) acc += rv * rv;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
}

// z = invd .* r ; out += r.z
__global__ __launch_bounds__(256) void emba_cg_z_kernel(const double* __restrict__ invd, const double* __restrict__ r, long N, double* __restrict__ z,
                                                        double* __restrict__ out, long sum_from = 0)
{
    __shared__ double s_w[4];
    double acc = 0.0;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < N; k += (long)gridDim.x * 256) { const double zv = invd[k] * r[k]; z[k] = zv; if (k >= sum_from) acc += r[k] * zv; }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
}

__global__ void emba_cg_p_kernel(double beta, const double* __restrict__ z, long N, double* __restrict__ pv)
{
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < N) pv[k] = z[k] + beta * pv[k];
}


// ---- sharded Schur solve: re-distribution of the records by pixel owner --------------------------------------------------------
// A pixel's two A12 columns are sums over ALL ranks' records of that pixel (its events come from several time shards), and the
// Schur complement needs the outer product of the SUMMED columns, so the records are first sent to the rank that owns their pixel:
// rank r of n owns the active pixels [P r / n, P (r+1) / n) in compact (ascending panorama) order.
__device__ __forceinline__ int pixel_owner(long k, long P, int n_ranks)
{
    int r = (int)((k * n_ranks) / (P > 0 ? P : 1));
    if (r >= n_ranks) r = n_ranks - 1;
    while (r + 1 < n_ranks && (P * (r + 1)) / n_ranks <= k) ++r;     // boundaries are floor(P r / n)
    while (r > 0 && (P * r) / n_ranks > k) --r;
    return r;
}

// (Round 5 had wave-aggregated tickets here — one atomic per owner and wave instead of one per record: 2-rank solve 11.6 -> 2.0 ms at the BASELINE shape.)
// Round 6: BLOCK-aggregated.  Tickets per wave still meant one RETURNING atomic per owner for every 8 records of the pack kernel (eight lanes per record), on
// n_ranks addresses of one line: 440 k of them at config 2's shape = 14.2 ms per launch (profiles/r06_two_rank_kernel_stats.txt) — returning atomics on one line are
// served one after the other, ~30 ns each.  Now a block counts its records per owner in LDS and takes ONE global ticket per owner (count kernel: one plain add).
constexpr int kShardMaxRanks = 1024;      // (emba_solve_shard_count refuses more)
constexpr int kShardPackRec = 2048;       // records per block of the pack kernel

__global__ __launch_bounds__(256) void emba_shard_count_kernel(RecView v, long n_rec, long P, int n_ranks, unsigned long long* __restrict__ cnt)
{
    __shared__ uint32_t s_cnt[kShardMaxRanks];
    for (int r = threadIdx.x; r < n_ranks; r += 256) s_cnt[r] = 0u;
    __syncthreads();
    for (long s = (long)blockIdx.x * 256 + threadIdx.x; s < n_rec; s += (long)gridDim.x * 256) {
        int32_t k = 0;
        if (rec_pixel(v, s, k)) atomicAdd(&s_cnt[pixel_owner(k, P, n_ranks)], 1u);
    }
    __syncthreads();
    for (int r = threadIdx.x; r < n_ranks; r += 256) if (s_cnt[r]) atomicAdd(cnt + r, (unsigned long long)s_cnt[r]);
}

// packed record = the record with its tail word rewritten to {compact pixel index, control-pose pair key}.  A block takes kShardPackRec consecutive slots: a first
// sweep over their tail words counts per owner (LDS), one ticket per owner gives the block its places in every owner's segment, a second sweep copies the records
// there — eight lanes per 128-B record (coalesced reads in slot order, one full line written per record), the place inside the block's run from an LDS cursor.
__global__ __launch_bounds__(256) void emba_shard_pack_kernel(RecView v, long n_rec, long P, int n_ranks, const unsigned long long* __restrict__ off,
                                                              unsigned long long* __restrict__ cursor, double* __restrict__ out)
{
    __shared__ uint32_t s_cnt[kShardMaxRanks];
    __shared__ unsigned long long s_base[kShardMaxRanks];
    const long s0 = (long)blockIdx.x * kShardPackRec, s1 = s0 + kShardPackRec < n_rec ? s0 + kShardPackRec : n_rec;
    for (int r = threadIdx.x; r < n_ranks; r += 256) s_cnt[r] = 0u;
    __syncthreads();
    for (long s = s0 + threadIdx.x; s < s1; s += 256) {
        int32_t k = 0;
        if (rec_pixel(v, s, k)) atomicAdd(&s_cnt[pixel_owner(k, P, n_ranks)], 1u);
    }
    __syncthreads();
    for (int r = threadIdx.x; r < n_ranks; r += 256) {
        const uint32_t n = s_cnt[r];
        s_base[r] = off[r] + (n ? atomicAdd(cursor + r, (unsigned long long)n) : 0ull);
        s_cnt[r] = 0u;                                               // ... becomes the cursor inside the block's run
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, c8 = threadIdx.x & 7;
    for (long sb = s0; sb < s1; sb += 32) {                          // (block-uniform trip count; every lane stays for the shuffles)
        const long s = sb + (threadIdx.x >> 3);
        const long sc = s < s1 ? s : s1 - 1;
        const double2 val = reinterpret_cast<const double2*>(v.rec + (size_t)kRecStride * sc)[c8];
        int32_t k = -1; uint32_t key = 0;
        unsigned long long pos = 0;
        bool ok = false;
        if (c8 == 7 && s < s1) {                                     // the lane that holds the tail word
            if (v.packed) { k = (int32_t)((long)(uint32_t)__double2loint(val.y) - v.pix_base); ok = true; }
            else { uint32_t pi; ok = record_valid(val.y, v.stamp, pi); if (ok) { k = v.compact[pi]; ok = k >= 0; } }
            if (ok) {
                const int owner = pixel_owner(k, P, n_ranks);
                key = v.packed ? (uint32_t)__double2hiint(val.y) : v.slot_key[s];
                pos = s_base[owner] + atomicAdd(&s_cnt[owner], 1u);
            }
        }
        const int src = lane | 7;
        const int okb = __shfl(ok ? 1 : 0, src);
        k = __shfl(k, src); key = (uint32_t)__shfl((int)key, src);
        pos = ((unsigned long long)(uint32_t)__shfl((int)(pos >> 32), src) << 32) | (uint32_t)__shfl((int)(pos & 0xFFFFFFFFull), src);
        if (!okb) continue;
        double2 o = val;
        if (c8 == 7) o.y = __hiloint2double((int)key, (int)k);
        reinterpret_cast<double2*>(out + (size_t)kRecStride * pos)[c8] = o;
    }
}

// S_aug += [A11m, . ; b1^T, 0] (the replicated part, added once after the all-reduce of the partial Schur sums)
__global__ void emba_schur_add_a11_kernel(const double* __restrict__ A11, const double* __restrict__ b1, int n, double lambda, double* __restrict__ S, long lds)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)n * n) {
        const int r = (int)(i % n), c = (int)(i / n);
        const double a = A11[i];
        S[(size_t)lds * c + r] += (r == c) ? a + lambda * a : a;
    }
    if (i < n) S[(size_t)lds * i + n] += b1[i];
}

}  // namespace emba
