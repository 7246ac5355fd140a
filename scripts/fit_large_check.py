#!/usr/bin/env python3
"""Diagnostic (GPU box): end-to-end fits at D = 256 (multitask K = 4, direct) through the many-workgroup kernels: finite, symmetric, timed."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import uglad_amd
from uglad_amd.utils.prepare_data import get_data
rng = np.random.default_rng(5)
Xb, Pb = get_data(256, (0.1, 0.2), 1024, 4, eig_offset=1.0, rng=rng)
t = time.time()
est = uglad_amd.uGLAD_multitask()
est.fit(list(Xb), epochs=20, lr=0.002, L=30, verbose=False)
torch.cuda.synchronize()
P = np.asarray(est.precision_)
print("multitask K=4 D=256 L=30, 20 epochs:", round(time.time() - t, 2), "s; precision_", P.shape, "finite", np.isfinite(P).all(), "symmetric", np.allclose(P, P.transpose(0, 2, 1)))
t = time.time()
g = uglad_amd.uGLAD_GL()
g.fit(Xb[0], epochs=20, lr=0.002, L=30, verbose=False)
print("direct D=256 L=30, 20 epochs:", round(time.time() - t, 2), "s; finite", np.isfinite(g.precision_).all())
