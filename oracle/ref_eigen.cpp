// oracle/ref_eigen.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" driver around the REFERENCE's own vendored Eigen (3.3.9, thirdparty/basalt-headers/thirdparty/eigen), compiled
// from where it lies under /root/reference, for the three library calls the reference's solvers make — so that the oracle's
// restatements of them (oracle/emba_oracle.c: ldlt_pivoted_solve, inverse2, the CG loop) are PINNED to what the reference executes:
//   S.ldlt().solve(rhs)                          src/emba/model.cpp:789   (Eigen/src/Cholesky/LDLT.h: pivoted, D pseudo-inverted)
//   A22m_i.inverse()  (Eigen::Matrix2d)          src/emba/model.cpp:750   (Eigen/src/LU/InverseImpl.h: 2x2 by the determinant)
//   Eigen::ConjugateGradient<SpMat, Lower|Upper> src/emba/model.cpp:828-836 (max 100 iterations, tolerance 1e-6, diagonal preconditioner)
// Built by oracle/Makefile into oracle/_ref/libref_eigen.so (git-ignored).  tests/golden/make_eigen_golden.py runs it on matrices the
// oracle produces and commits inputs + outputs as tests/golden/eigen_solvers.npz.
#include <Eigen/Dense>
#include <Eigen/IterativeLinearSolvers>
#include <Eigen/Sparse>
#include <cstdint>
#include <vector>

#include "utils/eigen_utils.h"     // the REFERENCE's include/utils/eigen_utils.h; its src/utils/eigen_utils.cpp is compiled next to this file (oracle/Makefile)

typedef Eigen::VectorXd VecXd;                 // include/emba/model.h:17-23
typedef Eigen::MatrixXd MatXd;
typedef Eigen::Matrix2d Mat2d;
typedef Eigen::SparseMatrix<double> SpMat;
typedef Eigen::Triplet<double> Triplet;

extern "C" {

// x = S.ldlt().solve(rhs).  S: n x n column-major (only its lower triangle is read by LDLT<MatXd, Lower>).
// vecD (n) and transp (n) receive the factorisation's D and transpositions (diagnostics of the pivoting); returns ldlt.info() != Success.
int ref_ldlt_solve(int n, const double* S, const double* rhs, double* x, double* vecD, int* transp)
{
    const Eigen::Map<const MatXd> Sm(S, n, n);
    const Eigen::Map<const VecXd> b(rhs, n);
    const MatXd Sc = Sm;
    Eigen::LDLT<MatXd> f = Sc.ldlt();
    const VecXd xs = f.solve(b);
    for (int i = 0; i < n; ++i) x[i] = xs(i);
    if (vecD) { const VecXd d = f.vectorD(); for (int i = 0; i < n; ++i) vecD[i] = d(i); }
    if (transp) for (int i = 0; i < n; ++i) transp[i] = (int)f.transpositionsP().indices()(i);
    return f.info() == Eigen::Success ? 0 : 1;
}

// out = A.inverse() for a 2x2 (row-major in and out; symmetric in the reference's use)
void ref_inverse2(const double* A, double* out)
{
    Mat2d M;
    M << A[0], A[1], A[2], A[3];
    const Mat2d I = M.inverse();
    out[0] = I(0, 0); out[1] = I(0, 1); out[2] = I(1, 0); out[3] = I(1, 1);
}

// x = cg.solve(b) for the n x n sparse matrix given by triplets (summed like setFromTriplets does), with the reference's settings.
void ref_cg_solve(int n, long nnz, const int32_t* rows, const int32_t* cols, const double* vals, const double* b, int max_iter, double tol,
                  double* x, int* iterations, double* error)
{
    std::vector<Triplet> t;
    t.reserve((size_t)nnz);
    for (long k = 0; k < nnz; ++k) t.emplace_back(rows[k], cols[k], vals[k]);
    SpMat A(n, n);
    A.setFromTriplets(t.begin(), t.end());
    A.makeCompressed();
    Eigen::ConjugateGradient<SpMat, Eigen::Lower | Eigen::Upper> cg;
    cg.setMaxIterations(max_iter);
    cg.setTolerance(tol);
    cg.compute(A);
    const Eigen::Map<const VecXd> bv(b, n);
    const VecXd xs = cg.solve(bv);
    for (int i = 0; i < n; ++i) x[i] = xs(i);
    *iterations = (int)cg.iterations();
    *error = cg.error();
}

// LEGM::solveNormalEqCG (src/emba/model.cpp:794-840) END TO END on the blocks formNormalEq + applyL2Reg export: the LM terms through the
// reference's own eigen_utils::diagMat / diagSpMat, the system [A11m A12; A12^T A22m] through its own eigen_utils::catSpMat (three calls,
// :819-821) — the functions of src/utils/eigen_utils.cpp, compiled unmodified — and A22 from its blocks as recoverA22FromBlocks does
// (:842-861, a LEGM member in model.cpp, which needs ROS/OpenCV headers: restated here, four triplets per pixel in its order).
// A11: n x n, A12: n x 2P (both column-major, dense like the reference's MatXd), A22_blocks: P x 4 row-major, b1: n, b2: 2P.
// Also returns the assembled matrix as triplets (nnz_out entries, capacity nnz_cap) so that the Python-side assembly can be cross-checked.
void ref_solve_normal_eq_cg(int n, long P, const double* A11p, const double* A12p, const double* A22_blocks, const double* b1p, const double* b2p,
                            double lambda, double* x1, double* x2, int* iterations, double* error,
                            long nnz_cap, int32_t* rows_out, int32_t* cols_out, double* vals_out, long* nnz_out)
{
    const Eigen::Map<const MatXd> A11(A11p, n, n);
    const Eigen::Map<const MatXd> A12m(A12p, n, 2 * P);
    const MatXd A12 = A12m;
    const Eigen::Map<const VecXd> b1(b1p, n), b2(b2p, 2 * P);
    const size_t dim_poses = (size_t)n, dim_map = (size_t)(2 * P);
    // recoverA22FromBlocks, model.cpp:842-861
    std::vector<Triplet> nz;
    nz.reserve(4 * (size_t)P);
    for (long i = 0; i < P; ++i) {
        const double* B = A22_blocks + 4 * i;
        nz.push_back(Triplet(2 * i, 2 * i, B[0]));
        nz.push_back(Triplet(2 * i + 1, 2 * i + 1, B[3]));
        nz.push_back(Triplet(2 * i + 1, 2 * i, B[2]));
        nz.push_back(Triplet(2 * i, 2 * i + 1, B[1]));
    }
    SpMat A22(2 * P, 2 * P);
    A22.setFromTriplets(nz.begin(), nz.end());
    A22.makeCompressed();
    // model.cpp:805-813
    const VecXd diag_A11 = A11.diagonal();
    MatXd D11 = eigen_utils::diagMat(diag_A11);
    const MatXd A11m = A11 + lambda * D11;
    const VecXd diag_A22 = A22.diagonal();
    SpMat D22 = eigen_utils::diagSpMat(diag_A22);
    SpMat A22m = A22 + lambda * D22;
    // :816-821
    const size_t dim_total = dim_poses + dim_map;
    VecXd b(dim_total);
    b << b1, b2;
    SpMat At, Ab, A;
    eigen_utils::catSpMat(2, A11m.sparseView(), A12.sparseView(), At);
    eigen_utils::catSpMat(2, A12.sparseView().transpose(), A22m, Ab);
    eigen_utils::catSpMat(1, At, Ab, A);
    // :824-831
    Eigen::ConjugateGradient<SpMat, Eigen::Lower | Eigen::Upper> cg;
    const int max_iter = 100;
    const double tol = 1e-6;
    cg.setMaxIterations(max_iter);
    cg.setTolerance(tol);
    cg.compute(A);
    VecXd x = cg.solve(b);
    for (size_t i = 0; i < dim_poses; ++i) x1[i] = x(i);
    for (size_t i = 0; i < dim_map; ++i) x2[i] = x(dim_poses + i);
    *iterations = (int)cg.iterations();
    *error = cg.error();
    long k = 0;
    for (int c = 0; c < A.outerSize(); ++c)
        for (SpMat::InnerIterator it(A, c); it; ++it, ++k)
            if (k < nnz_cap) { rows_out[k] = (int32_t)it.row(); cols_out[k] = (int32_t)it.col(); vals_out[k] = it.value(); }
    *nnz_out = k;
}

}  // extern "C"
