"""TEST INFRASTRUCTURE (the checker, never the product): CPU restatement in fp64 numpy of the table -> covariance front-end
of the reference, pinned by tests/golden/cov_*.npz (outputs of the real reference, tests/golden/make_cov_goldens.py).

  normalize_min_max   /root/reference/uglad/utils/prepare_data.py:597-613 (normalize_table, typeN="min_max")
  empirical_cov       sklearn.covariance.empirical_covariance(assume_centered=False), the call at prepare_data.py:342-344
  get_covariance      prepare_data.py:328-356: where min eig <= 1e-6, S += (offset - min eig) I
"""
import numpy as np


def normalize_min_max(X):
    X = np.asarray(X, dtype=np.float64)
    mn, mx = X.min(axis=-2, keepdims=True), X.max(axis=-2, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (X - mn) / (mx - mn)


def empirical_cov(X):
    X = np.asarray(X, dtype=np.float64)
    Xc = X - X.mean(axis=-2, keepdims=True)
    return np.swapaxes(Xc, -1, -2) @ Xc / X.shape[-2]


def get_covariance(Xb, offset=0.1):
    out = []
    for X in np.asarray(Xb, dtype=np.float64):
        S = empirical_cov(X)
        eig = np.linalg.eigvals(S).real  # (the reference takes the real parts of the general eigenvalue routine)
        if eig.min() <= 1e-6:
            S = S + np.eye(S.shape[-1]) * (offset - eig.min())
        out.append(S)
    return np.array(out)
