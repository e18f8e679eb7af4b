"""the Poisson-reconstruction oracle (oracle/poisson.py) pinned by the equation it solves."""
import numpy as np

from oracle import poisson as OP


def laplacian_zero_boundary(M):
    P = np.pad(M, 1)
    return P[2:, 1:-1] + P[:-2, 1:-1] + P[1:-1, 2:] + P[1:-1, :-2] - 4.0 * M


def test_dst_definition_matches_scipy():
    X = np.random.default_rng(0).normal(size=(13, 22))
    from scipy import fft as sfft
    assert np.allclose(OP.dst1_dense(X), sfft.dstn(X, type=1), rtol=1e-12, atol=1e-12)
    # RODFT00 applied twice = 2(n+1) per dimension (laplace.cpp:645 fft_norm)
    assert np.allclose(OP.dst1_dense(OP.dst1_dense(X)), 4.0 * 14 * 23 * X, rtol=1e-12, atol=1e-10)


def test_oracle_solves_the_discrete_poisson_equation():
    rng = np.random.default_rng(1)
    for shape in [(8, 16), (33, 66), (128, 256)]:
        Gx, Gy = rng.normal(size=shape), rng.normal(size=shape)
        F = OP.divergence(Gx, Gy)
        assert (F[-1, :] == 0).all() and (F[:, -1] == 0).all()
        i, j = 2, 3
        assert F[i, j] == Gx[i, j + 1] - Gx[i, j] + Gy[i + 1, j] - Gy[i, j]
        M = OP.reconstruct_from_gradient(Gx, Gy)
        assert np.abs(laplacian_zero_boundary(M) - F).max() < 1e-10 * max(1.0, np.abs(F).max()) * shape[1]


def test_gradient_of_a_smooth_image_is_inverted():
    # the gradient field of an image that vanishes on and beyond the border integrates back to that image
    H, W = 64, 128
    y, x = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    L = np.sin(np.pi * (x + 1) / (W + 1)) * np.sin(2 * np.pi * (y + 1) / (H + 1))
    Lp = np.pad(L, ((0, 1), (0, 1)))                   # value 0 outside (Dirichlet)
    Gx = np.zeros((H, W)); Gy = np.zeros((H, W))
    Gx[:, 1:] = L[:, 1:] - L[:, :-1]; Gx[:, 0] = L[:, 0]          # backward differences so that forward div = 5-point Laplacian
    Gy[1:, :] = L[1:, :] - L[:-1, :]; Gy[0, :] = L[0, :]
    M = OP.reconstruct_from_gradient(Gx, Gy)
    # interior rows/cols (the reference leaves the last row/column of F at zero): compare away from that edge
    assert np.abs(M - L)[: H // 2, : W // 2].max() < 0.05
