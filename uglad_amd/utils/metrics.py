"""Support-recovery metrics reported by ``fit`` when a true precision matrix is supplied.

Mirrors the dictionary produced by the reference's ``report_metrics_all``
(``uglad/utils/metrics.py:25-108``): edges are the strict upper triangle, an edge is
"predicted" when the entry is non-zero, scores for AUC/AUPR are |entry|.  ROC-AUC and
average precision are computed here with numpy (trapezoid over distinct thresholds /
step-wise precision-recall sum, the definitions sklearn uses) so the GPU box needs no
sklearn for the parity report.
"""
from __future__ import annotations

import numpy as np

__all__ = ["get_auc", "report_metrics_all", "report_metrics", "summarize_compare_theta"]


def _curve_counts(y: np.ndarray, scores: np.ndarray):
    order = np.argsort(-scores, kind="mergesort")
    y = y[order]
    s = scores[order]
    distinct = np.where(np.diff(s))[0]
    idx = np.r_[distinct, y.size - 1]
    tps = np.cumsum(y)[idx].astype(np.float64)
    fps = (1 + idx - tps).astype(np.float64)
    return tps, fps


def get_auc(y, scores):
    """(ROC-AUC, average precision) of binary labels ``y`` against ``scores`` (ref :7-22)."""
    y = np.asarray(y).astype(int)
    scores = np.asarray(scores, dtype=np.float64)
    P = float(y.sum())
    N = float(y.size - P)
    if y.size == 0 or P == 0 or N == 0:
        return float("nan"), float("nan")
    tps, fps = _curve_counts(y, scores)
    tpr = np.r_[0.0, tps / P]
    fpr = np.r_[0.0, fps / N]
    auc = float(np.trapezoid(tpr, fpr)) if hasattr(np, "trapezoid") else float(np.trapz(tpr, fpr))
    precision = tps / (tps + fps)
    recall = tps / P
    aupr = float(np.sum(np.diff(np.r_[0.0, recall]) * precision))
    return auc, aupr


def _edges(trueG, G):
    trueG = np.asarray(trueG).real
    G = np.asarray(G).real
    iu = np.triu_indices(G.shape[-1], 1)
    t = (trueG[iu] != 0).astype(int)
    p = (G[iu] != 0).astype(int)
    return t, p, np.abs(G[iu])


def report_metrics_all(trueG, G, beta: int = 1) -> dict:
    """FDR/TPR/FPR/SHD/nnz/precision/recall/F-beta/AUPR/AUC, rounded to 3 decimals (ref :25-108)."""
    t, p, score = _edges(trueG, G)
    auc, aupr = get_auc(t, score)
    TP = float(np.sum(t * p))
    mism = np.logical_xor(t, p)
    FP = float(np.sum(mism * p))
    FN = float(np.sum(mism * t))
    P = float(p.sum())
    T = float(t.sum())
    F = float(t.size - T)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = {
            "FDR": np.float64(FP) / P,
            "TPR": np.float64(TP) / T,
            "FPR": np.float64(FP) / F,
            "SHD": float(mism.sum()),
            "nnzTrue": T,
            "nnzPred": P,
            "precision": np.float64(TP) / (TP + FP),
            "recall": np.float64(TP) / (TP + FN),
            "Fbeta": np.float64((1 + beta**2) * TP) / ((1 + beta**2) * TP + beta**2 * FN + FP),
            "aupr": aupr,
            "auc": auc,
        }
    return {k: round(float(v), 3) for k, v in out.items()}


def report_metrics(trueG, G, beta: int = 1) -> dict:
    """The three-number variant (ref :148-191), unrounded."""
    t, p, score = _edges(trueG, G)
    auc, aupr = get_auc(t, score)
    TP = float(np.sum(t * p))
    mism = np.logical_xor(t, p)
    FP = float(np.sum(mism * p))
    FN = float(np.sum(mism * t))
    with np.errstate(divide="ignore", invalid="ignore"):
        fb = np.float64((1 + beta**2) * TP) / ((1 + beta**2) * TP + beta**2 * FN + FP)
    return {"Fbeta": float(fb), "aupr": aupr, "auc": auc}


def summarize_compare_theta(compare_dict_list, method_name: str = "Method Name", verbose: bool = True) -> dict:
    """Mean and standard deviation of every metric over runs (ref :111-145)."""
    out = {}
    for key in compare_dict_list[0]:
        vals = [d[key] for d in compare_dict_list]
        out[key] = (round(float(np.mean(vals)), 3), round(float(np.std(vals)), 3))
    if verbose:
        print(f"Avg results for {method_name} (mean, std): {out}  [runs={len(compare_dict_list)}]")
    return out
