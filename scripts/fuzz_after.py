#!/usr/bin/env python3
"""Diagnostic (GPU box): random shapes through the entry points either side of the pass -- uglad_covariance (K tables of N x D, with and without
min-max normalisation, eigenvalue repair) against oracle/covariance.py, uglad_conditional_mean (random observed sets, incl. none-but-one and
all-but-one) against oracle/after_path.py, uglad_partial_correlations.   python scripts/fuzz_after.py [seed=0] [cases=120]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uglad_amd
from uglad_amd import _lib
from oracle import covariance as oc, after_path as oa

def relF(a, b): return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / max(np.linalg.norm(np.asarray(b, np.float64)), 1e-300))
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rng = np.random.default_rng(seed)
lib = _lib.get_lib()
bad = 0; w = {"cov": 0.0, "mean": 0.0, "ccov": 0.0, "logpdf": 0.0, "pcorr": 0.0}
for case in range(cases):
    D = int(rng.integers(1, 257)) if rng.random() < 0.3 else int(rng.integers(1, 49))
    K = int(rng.integers(1, 5)); N = int(rng.choice([1, 2, 3, max(2, D // 2), D, 2 * D + 1, 300]))
    norm = bool(rng.integers(0, 2)) and N >= 2
    X = rng.random((K, N, D)) * rng.choice([1.0, 10.0]) + rng.standard_normal((K, 1, D))
    if norm and (X.max(axis=1) == X.min(axis=1)).any(): norm = False
    X32 = np.ascontiguousarray(X.astype(np.float32))
    S = lib.covariance(torch.from_numpy(X32).cuda(), normalize=norm, eval_offset=0.1)
    Xo = oc.normalize_min_max(X32.astype(np.float64)) if norm else X32.astype(np.float64)
    ref = oc.get_covariance(Xo, 0.1)
    # the repair decision sits on min eig <= 1e-6: a covariance whose smallest eigenvalue is within fp32 noise of it may be repaired on one side only
    mins = np.array([np.linalg.eigvalsh(oc.empirical_cov(x)).min() for x in Xo])
    scale = np.array([np.abs(oc.empirical_cov(x)).max() for x in Xo])
    border = np.abs(mins - 1e-6) < 2e-6 * np.maximum(scale, 1.0)
    e = max((relF(S[k].cpu().numpy(), ref[k]) for k in range(K) if not border[k]), default=0.0)
    w["cov"] = max(w["cov"], e)
    tag = f"case {case}: K={K} N={N} D={D} normalize={norm}"
    if e > 2e-5: bad += 1; print(tag, f"covariance {e:.2e}   <--", flush=True)
    # conditional Gaussian on a well-conditioned precision of the same size
    A = rng.standard_normal((K, D, D)); P = A @ A.transpose(0, 2, 1) / D + np.eye(D)[None] * 0.5
    mu = rng.random((K, D)); vals = rng.random((K, D))
    nobs = int(rng.choice([1, max(1, D // 3), max(1, D - 1)])) if D > 1 else 0
    if D > 1:
        mask = np.zeros((K, D), np.float32)
        for k in range(K): mask[k, rng.choice(D, nobs, replace=False)] = 1.0
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).cuda()
        fm, cc, lp = lib.conditional_mean(f32(P), f32(mu), f32(mask), f32(vals))
        for k in range(K):
            obs = np.nonzero(mask[k])[0]; un = np.nonzero(mask[k] == 0)[0]
            P32, mu32, v32 = P[k].astype(np.float32).astype(np.float64), mu[k].astype(np.float32).astype(np.float64), vals[k].astype(np.float32).astype(np.float64)
            full, cov, lpdf = oa.conditional_gaussian(P32, mu32, obs, v32[obs])
            e1 = relF(fm[k].cpu().numpy(), full); e2 = relF(cc[k].cpu().numpy()[np.ix_(un, un)], cov); e3 = abs(lp[k].item() - lpdf) / max(abs(lpdf), 1.0)
            w["mean"] = max(w["mean"], e1); w["ccov"] = max(w["ccov"], e2); w["logpdf"] = max(w["logpdf"], e3)
            if e1 > 2e-5 or e2 > 5e-5 or e3 > 2e-5: bad += 1; print(tag, f"observed {nobs}: mean {e1:.2e} cov {e2:.2e} log pdf {e3:.2e}   <--", flush=True)
    rho = lib.partial_correlations(torch.from_numpy(np.ascontiguousarray(P.astype(np.float32))).cuda())
    e4 = max(relF(rho[k].cpu().numpy(), oa.partial_correlations(P[k].astype(np.float32))) for k in range(K))
    w["pcorr"] = max(w["pcorr"], e4)
    if e4 > 5e-6: bad += 1; print(tag, f"partial correlations {e4:.2e}   <--", flush=True)
print(f"seed {seed}: {cases} cases; worst covariance {w['cov']:.2e}, conditional mean {w['mean']:.2e}, conditional covariance {w['ccov']:.2e}, log pdf {w['logpdf']:.2e}, partial correlations {w['pcorr']:.2e}; flagged {bad}")
