"""ORACLE (test infrastructure, not product code) -- fp64 numpy restatement of what follows the hot path in the reference
(SURVEY.md 8f N3, N4).  Only tests/ may import this file; the shipped package `uglad_amd` never does.

  conditional_gaussian   uglad/main.py:1176-1227  conditional_gaussian_with_probabilities: partition the precision matrix into
                         unobserved / observed coordinates, conditional mean mean_u - L_uu^-1 L_uo (x_o - mean_o), conditional
                         covariance L_uu^-1, density of N(map; map, L_uu^-1) = (2 pi)^(-n_u/2) det(L_uu)^(1/2)
  map_estimate           uglad/main.py:1229-1260  compute_map_estimate: the full mean vector clipped to [0, 1]
  partial_correlations   uglad/main.py:796-821    get_partial_correlations: -p_ij / sqrt(p_ii p_jj) from the upper triangle, mirrored
  support_metrics        uglad/utils/metrics.py:25-108 report_metrics_all: counts on the strict upper triangle + the ranking
                         metrics by their definitions (Mann-Whitney with ties at 1/2; average precision over distinct scores)

Parity pin: tests/test_after_path.py checks every function against tests/golden/map_*.npz and metrics_k3_d20.npz, which
tests/golden/make_goldens_r2.py captured from the real reference in the build container.
"""
from __future__ import annotations

import numpy as np

METRIC_KEYS = ("FDR", "TPR", "FPR", "SHD", "nnzTrue", "nnzPred", "precision", "recall", "Fbeta", "aupr", "auc")


def conditional_gaussian(precision, mean, observed_idx, observed_values):
    P = np.asarray(precision, dtype=np.float64)
    mu = np.asarray(mean, dtype=np.float64)
    n = mu.shape[0]
    obs = [int(i) for i in observed_idx]
    un = [i for i in range(n) if i not in set(obs)]
    x = np.asarray(observed_values, dtype=np.float64)
    L11, L12 = P[np.ix_(un, un)], P[np.ix_(un, obs)]
    cond = mu[un] - np.linalg.solve(L11, L12 @ (x - mu[obs]))
    cov = np.linalg.inv(L11)
    full = np.zeros(n)
    full[un] = cond
    full[obs] = x
    sign, lad = np.linalg.slogdet(L11)
    log_pdf = -0.5 * len(un) * np.log(2.0 * np.pi) + 0.5 * lad if sign > 0 else np.nan
    return full, cov, log_pdf


def map_estimate(precision, mean, observed_idx, observed_values):
    return np.clip(conditional_gaussian(precision, mean, observed_idx, observed_values)[0], 0.0, 1.0)


def partial_correlations(precision):
    P = np.asarray(precision, dtype=np.float64)
    D = P.shape[0]
    rho = np.eye(D)
    for i in range(D):
        for j in range(i + 1, D):
            rho[i, j] = rho[j, i] = -P[i, j] / np.sqrt(P[i, i] * P[j, j])
    return rho


def support_metrics(true_theta, pred_theta, beta: int = 1):
    """The 11 numbers of report_metrics_all, unrounded, in METRIC_KEYS order (pure-Python loops: small cases only)."""
    T = np.asarray(true_theta).real
    G = np.asarray(pred_theta).real
    iu = np.triu_indices(G.shape[-1], 1)
    t = T[iu] != 0
    p = G[iu] != 0
    s = np.abs(G[iu])
    TP = float(np.sum(t & p))
    FP = float(np.sum(~t & p))
    FN = float(np.sum(t & ~p))
    nT, nP = float(t.sum()), float(p.sum())
    nF = float(t.size) - nT
    mw2, ap = 0, 0.0
    for i in np.nonzero(t)[0]:
        mw2 += 2 * int(np.sum(~t & (s < s[i]))) + int(np.sum(~t & (s == s[i])))
        ap += float(np.sum(t & (s >= s[i]))) / float(np.sum(s >= s[i]))
    with np.errstate(divide="ignore", invalid="ignore"):
        b2 = float(beta) ** 2
        return np.array([np.float64(FP) / nP, np.float64(TP) / nT, np.float64(FP) / nF, FP + FN, nT, nP,
                         np.float64(TP) / (TP + FP), np.float64(TP) / (TP + FN),
                         np.float64((1 + b2) * TP) / ((1 + b2) * TP + b2 * FN + FP),
                         ap / nT if nT > 0 and nF > 0 else np.nan,
                         mw2 / (2.0 * nT * nF) if nT > 0 and nF > 0 else np.nan])
