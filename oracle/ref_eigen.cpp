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

}  // extern "C"
