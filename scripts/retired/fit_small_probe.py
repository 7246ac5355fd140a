#!/usr/bin/env python3
"""RETIRED (round 4): ran against commit 60c0897^ and earlier, when the library still had its hipGraph cache (uglad_graph_cache_stats, UGLAD_GRAPHS);
that cache was deleted in round 3 on this script's own evidence (profiles/r03_fit_small_graph_probe.txt), so it no longer runs against HEAD.

Diagnostic (GPU box): the plain `uGLAD_GL().fit(X)` call on a small problem (BASELINE config 1: D = 25, L = 15) -- wall time per epoch
and the hipGraph cache counters, with graphs on and off (UGLAD_GRAPHS=0)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uglad_amd
from uglad_amd import _lib
g = np.load(os.path.join(ROOT, "tests", "golden", "fit_direct_d25.npz"))
X = g["X"]
for mode, kw in (("direct", {}), ("cv", dict(k_fold=3)), ("cv batched", dict(k_fold=3, batched_folds=True))):
    est = uglad_amd.uGLAD_GL()
    est.fit(X.copy(), epochs=20, lr=0.002, L=15, verbose=False, mode=mode.split()[0], **kw)  # warm-up (captures)
    torch.cuda.synchronize()
    t0 = time.time()
    est.fit(X.copy(), epochs=200, lr=0.002, L=15, verbose=False, mode=mode.split()[0], **kw)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"UGLAD_GRAPHS={os.environ.get('UGLAD_GRAPHS', '1')} fit({mode}) D=25 L=15: {dt / 200 * 1e3:.3f} ms per epoch; graph cache {_lib.get_lib().graph_cache_stats()}")
