#!/usr/bin/env python3
"""Timings of the BASELINE.json configurations on one MI355X (GPU box).  Prints one line per configuration."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uglad_amd
from uglad_amd import main as um
from uglad_amd.dist import Collective
from uglad_amd.utils.prepare_data import synthetic_covariance_batch, get_data

pz = np.load(os.path.join(ROOT, "tests", "golden", "params_trained.npz"))
def model():
    m = uglad_amd.GladParams(1.0, device="cuda")
    m.load_state_dict({k: torch.from_numpy(np.array(pz[k])) for k in pz.files})
    return m
def sync(): torch.cuda.synchronize()
def time_steps(S, L, steps=5, loss_S=None):
    m = model(); opt = uglad_amd.get_optimizers(m)
    one = Collective()
    def step():
        opt.zero_grad()
        th, ls = um.forward_uGLAD(S, m, L=L, collective=one, loss_Sb=loss_S)
        ls.backward(); opt.step()
    step(); sync()
    t = time.perf_counter()
    for _ in range(steps): step()
    sync(); tt = (time.perf_counter() - t) / steps
    with torch.no_grad():
        um.forward_uGLAD(S, m, L=L, collective=one, loss_Sb=loss_S); sync()
        t = time.perf_counter()
        for _ in range(steps): um.forward_uGLAD(S, m, L=L, collective=one, loss_Sb=loss_S)
        sync(); tf = (time.perf_counter() - t) / steps
    return tt, tf
rows = []
for name, M, D, L in (("C1 demo single graph", 1, 25, 15), ("C2", 128, 64, 30), ("C3", 1024, 128, 30)):
    base = synthetic_covariance_batch(min(M, 32), D, seed=11)
    S = torch.from_numpy(np.tile(base, (M // base.shape[0] + 1, 1, 1))[:M].copy()).cuda()
    tt, tf = time_steps(S, L)
    print(f"{name}: M={M} D={D} L={L}: train {tt*1e3:8.2f} ms/pass = {M*L/tt:10.0f} unroll-steps/s ; forward-only {tf*1e3:8.2f} ms = {M*L/tf:10.0f} unroll-steps/s", flush=True)
# C5 shape on one GPU: K=8 sub-sample covariances D=256, loss vs one full covariance
base = synthetic_covariance_batch(9, 256, seed=12)
S_K = torch.from_numpy(base[:8]).cuda(); S_full = torch.from_numpy(base[8:9]).cuda()
tt, tf = time_steps(S_K, 30, steps=3, loss_S=S_full)
print(f"C5 shape (one GPU): K=8 D=256 L=30: train {tt*1e3:8.2f} ms/pass = {8*30/tt:10.0f} unroll-steps/s ; forward-only {tf*1e3:8.2f} ms = {8*30/tf:10.0f} unroll-steps/s", flush=True)
# C1 end to end: uGLAD_GL.fit(direct) as in the demo notebook (D=25 here, 100 epochs)
X, _ = get_data(25, (0.1, 0.2), 500, 1, eig_offset=1.0, rng=3)
est = uglad_amd.uGLAD_GL()
est.fit(X[0], epochs=10, L=15, verbose=False); sync()
t = time.perf_counter(); est.fit(X[0], epochs=100, L=15, verbose=False); sync(); tfit = time.perf_counter() - t
print(f"C1 fit(direct) D=25 L=15 100 epochs: {tfit:.3f} s total = {100*15/tfit:.0f} unroll-steps/s (reference on 8 CPU threads here: 3.9 s, 385/s)")
# Covariance front-end: K tables of N x D on the device vs the host path fit() uses by default
from uglad_amd import _lib
from uglad_amd.utils import prepare_data
lib = _lib.get_lib()
for K, N, D in ((1024, 500, 128), (8, 1024, 256)):
    Xh = np.random.default_rng(5).random((K, N, D)).astype(np.float32)
    Xd = torch.from_numpy(Xh).cuda()
    lib.covariance(Xd, normalize=True); sync()
    tds = []  # the median of single calls: an average carries the one call that pays a hipMalloc when a scratch size misses the allocator's pool (profiles/r04_cov_probe.txt)
    for _ in range(7):
        t = time.perf_counter(); lib.covariance(Xd, normalize=True); sync(); tds.append(time.perf_counter() - t)
    td = float(np.median(tds))
    t = time.perf_counter()
    kk = min(K, 32)
    prepare_data.get_covariance(np.stack([np.array(prepare_data.normalize_table(__import__("pandas").DataFrame(x.astype(np.float64)), "min_max")) for x in Xh[:kk]]))
    th = (time.perf_counter() - t) * K / kk
    print(f"covariance front-end K={K} N={N} D={D}: device (normalise + covariance + eigenvalue repair) {td*1e3:.2f} ms = {K*N*D*4/td/1e9:.0f} GB/s of table "
          f"read; host numpy path {th*1e3:.0f} ms (extrapolated from {kk} tables)", flush=True)
