#!/usr/bin/env python3
"""Diagnostic (GPU box): where the time of uglad_covariance goes at BASELINE config 5's shape (K = 8 tables of 1024 x 256) -- the
contraction alone (repair=False) vs with the eigenvalue repair, on uniform-random tables (covariance ~ I / 12: a tightly clustered
spectrum) and on tables sampled from a Gaussian graph (what fit() sees).  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uglad_amd import _lib
from uglad_amd.utils.prepare_data import get_data
lib = _lib.get_lib()
def sync(): torch.cuda.synchronize()
def t(fn, n=5):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    sync()
    return (time.perf_counter() - t0) / n * 1e3
shapes = [(8, 1024, 256), (1024, 500, 128), (8, 1024, 128), (64, 1024, 256)]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
for K, N, D in shapes:
    Xu = torch.from_numpy(np.random.default_rng(5).random((K, N, D)).astype(np.float32)).cuda()
    Xg, _ = get_data(D, (0.1, 0.2), N, min(K, 8), eig_offset=1.0, rng=7)
    Xg = torch.from_numpy(np.tile(np.stack(Xg).astype(np.float32), (K // min(K, 8) + 1, 1, 1))[:K].copy()).cuda()
    for name, X in (("uniform tables", Xu), ("Gaussian-graph tables", Xg)):
        a = t(lambda: lib.covariance(X, normalize=True, repair=False))
        b = t(lambda: lib.covariance(X, normalize=True, repair=True))
        S = lib.covariance(X, normalize=True, repair=False)
        U, be = torch.empty_like(S), torch.empty(K, D, device="cuda")
        wsp = lib.workspace(K, D, S)
        c = t(lambda: lib.symeig(S, U, be))
        ev = torch.linalg.eigvalsh(S.double())
        print(f"K={K} N={N} D={D} {name:22s}: contraction {a:7.3f} ms, with repair {b:7.3f} ms, symeig alone {c:7.3f} ms; "
              f"spectrum of S[0]: min {ev[0,0]:.3e} max {ev[0,-1]:.3e} smallest gap/max {(ev[0,1:]-ev[0,:-1]).min()/ev[0,-1]:.1e}", flush=True)
