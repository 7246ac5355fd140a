#!/usr/bin/env python3
"""Diagnostic (GPU box): where does the time of a SMALL pass (C1: one 25 x 25 matrix, L = 15) go -- the kernels themselves,
or the Python / autograd / optimiser around them?"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uglad_amd
from uglad_amd import _lib, main as um
from uglad_amd.utils.prepare_data import synthetic_covariance_batch
M, D, L = 1, 25, 15
lib = _lib.get_lib()
S = torch.from_numpy(synthetic_covariance_batch(M, D, seed=1)).cuda().contiguous()
model = uglad_amd.GladParams(1.0, device="cuda")
opt = uglad_amd.get_optimizers(model, lr_glad=0.002)
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def full():
    opt.zero_grad(); th, loss = um.forward_uGLAD(S, model, L=L); loss.backward(); opt.step()
def fwd_bwd():
    th, loss = um.forward_uGLAD(S, model, L=L); loss.backward()
def fwd_only():
    with torch.no_grad(): um.forward_uGLAD(S, model, L=L)
f32 = dict(dtype=torch.float32, device="cuda")
p = model.packed().detach()
Z = torch.empty(L + 1, M, D, D, **f32); half = torch.empty(L, M, D, D, **f32); U = torch.empty(L, M, D, D, **f32)
beta = torch.empty(L, M, D, **f32); lam = torch.empty(L + 1, **f32); lam_in = torch.empty(L + 1, 2, **f32)
nfp = torch.empty(M, **f32); nfs = torch.empty(1, **f32); wsp = lib.workspace(M, D, S)
g0, g1 = torch.empty(M, D, D, **f32), torch.empty(M, D, D, **f32)
grp, glp, gtp, grad = torch.zeros(M, 28, **f32), torch.empty(L, M, **f32), torch.empty(M, **f32), torch.empty(42, **f32)
G = torch.randn(M, D, D, **f32); G = (G + G.transpose(1, 2)).contiguous()
def c_fwd():
    lib.glad_forward(S, p, 1.0, 0, L, Z, half, U, beta, lam, lam_in, nfp, nfs, wsp, 1)
def c_bwd():
    lib.glad_backward(G, S, p, 0, L, Z, half, U, beta, lam, lam_in, g0, g1, grp, glp, gtp, grad, wsp, 1)
c_fwd()
print(f"C1 (M=1, D=25, L=15), ms per call: full training step {t(full):.3f} | forward+loss+backward {t(fwd_bwd):.3f} | "
      f"no_grad forward+loss {t(fwd_only):.3f} | C glad_forward alone {t(c_fwd):.3f} | C glad_backward alone {t(c_bwd):.3f}")
# ---- do passes on different streams overlap on the GPU?
NS = 4
streams = [torch.cuda.Stream() for _ in range(NS)]
bufs = []
for _ in range(NS):
    bufs.append(dict(Z=torch.empty_like(Z), half=torch.empty_like(half), U=torch.empty_like(U), beta=torch.empty_like(beta),
                     lam=torch.empty_like(lam), lam_in=torch.empty_like(lam_in), nfp=torch.empty_like(nfp), nfs=torch.empty_like(nfs),
                     wsp=torch.empty_like(wsp)))
def multi():
    for s, b in zip(streams, bufs):
        with torch.cuda.stream(s):
            lib.glad_forward(S, p, 1.0, 0, L, b["Z"], b["half"], b["U"], b["beta"], b["lam"], b["lam_in"], b["nfp"], b["nfs"], b["wsp"], 1)
torch.cuda.synchronize()
print(f"{NS} forward passes on {NS} streams, enqueued from one thread: {t(multi, 50):.3f} ms (one pass alone: {t(c_fwd):.3f} ms)")
