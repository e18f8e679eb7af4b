"""CPU oracle for SURVEY §8 f3 (test infrastructure only): the reference's Poisson reconstruction restated in numpy.

Follows poisson_reconstruction::reconstructFromGradient (reference src/image_rec/poisson_reconstruction.cpp:9-50) and
pde::poisolve, Dirichlet branch (src/image_rec/laplace.cpp:587-797) step by step.  The reference's transform is FFTW's
FFTW_RODFT00 (fftw3 is a system dependency, not under /root/reference, version unpinned): its published definition,
Y[k] = 2 * sum_j X[j] sin(pi (j+1)(k+1)/(n+1)), is what scipy.fft.dst(type=1) computes (pocketfft); parity unpinned with
respect to FFTW's own rounding.  The restatement is pinned mathematically instead: tests/test_poisson_cpu.py checks that the
result solves the discrete Poisson equation it is defined by, and against a dense-matrix DST for small sizes."""
import numpy as np
from scipy import fft as sfft


def divergence(Gx, Gy):
    """F = d gx/dx + d gy/dy, forward differences; last row and last column stay 0 (poisson_reconstruction.cpp:17-30)."""
    H, W = Gx.shape
    F = np.zeros((H, W))
    F[:H - 1, :W - 1] = Gx[:H - 1, 1:] - Gx[:H - 1, :W - 1] + Gy[1:, :W - 1] - Gy[:H - 1, :W - 1]
    return F


def poisolve_dirichlet(F):
    """laplace.cpp:587-797 with a1 = a2 = h1 = h2 = 1, zero boundary values, add_boundary_to_solution = false."""
    n1, n2 = F.shape
    rhs = sfft.dstn(F, type=1)                                   # :639-644, RODFT00 in both dimensions (factor 2 each)
    rhs = rhs * (1.0 / (4.0 * ((n1 + 1) * (n2 + 1))))            # :645, :679-683
    lam1 = -4.0 * np.sin((np.pi * (np.arange(n1) + 1)) / (2.0 * (n1 + 1))) ** 2      # :700-702
    lam2 = -4.0 * np.sin((np.pi * (np.arange(n2) + 1)) / (2.0 * (n2 + 1))) ** 2      # :703-704
    U = rhs / (lam1[:, None] + lam2[None, :])                     # :716-731 (div != 0 for Dirichlet)
    return sfft.dstn(U, type=1)                                   # :749-753


def reconstruct_from_gradient(Gx, Gy):
    return poisolve_dirichlet(divergence(np.asarray(Gx, dtype=np.float64), np.asarray(Gy, dtype=np.float64)))


def dst1_dense(X):
    """The same 2-D DST-I by explicit sine matrices (small sizes only): the definition, independent of any FFT."""
    n1, n2 = X.shape
    S1 = 2.0 * np.sin(np.pi * np.outer(np.arange(1, n1 + 1), np.arange(1, n1 + 1)) / (n1 + 1))
    S2 = 2.0 * np.sin(np.pi * np.outer(np.arange(1, n2 + 1), np.arange(1, n2 + 1)) / (n2 + 1))
    return S1 @ X @ S2
