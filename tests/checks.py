"""Shared assertions of the GPU parity tests."""
import numpy as np


def assert_normals_match(nrm, onrm, pts, nbr, k=20, tol=1e-9):
    """GPU and oracle normals agree to `tol` except where the PCA direction is not determined by the
    data: every mismatch must have a degenerate spectrum (the two smallest |eigenvalues| of the
    reference's moment matrix -- float32 products summed in double, em_icp.hpp:303-323 -- closer
    than 1e-6 of the largest), and those must be rare."""
    dots = np.abs(np.einsum("ni,ni->n", nrm, onrm))
    bad = np.nonzero(~(1 - dots < tol))[0]
    assert len(bad) <= 2e-3 * len(nrm), len(bad)
    if len(bad) == 0:
        return
    nb = nbr[bad]
    P = pts[np.maximum(nb, 0)].astype(np.float32)                  # [m, k, 3]
    live = (nb >= 0)[..., None]
    mean = np.where(live, P.astype(np.float64), 0.0).sum(axis=1) / k
    prod = (P[:, :, :, None] * P[:, :, None, :]).astype(np.float64)  # float32 products (quirk Q2)
    cov = np.where(live[..., None], prod, 0.0).sum(axis=1) / k - mean[:, :, None] * mean[:, None, :]
    lam = np.sort(np.abs(np.linalg.eigvalsh(cov)), axis=1)
    gap = (lam[:, 1] - lam[:, 0]) / np.maximum(lam[:, 2], 1e-300)
    assert (gap < 1e-6).all(), (len(bad), float(gap.max()))
