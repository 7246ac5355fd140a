#!/usr/bin/env python3
"""Diagnostic (GPU box): BASELINE config 1 (one 25 x 25 matrix, L = 15) -- parity with the golden and the median time of a training pass / a forward-only
pass, for the library UGLAD_LIB points at.  Round 4 used it to compare workgroup sizes of the NT = 1 kernels (dev builds with -DUGLAD_THREADS=128 / 256
-DUGLAD_MAX_NT=1: profiles/r04_c1_threads.txt -- no difference: the step is two dependent chains on a few waves, not barrier width)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import uglad_amd
from oracle import glad_exact as ex
from uglad_amd import _lib, main as um
from uglad_amd.dist import Collective
g = np.load("/root/repo/tests/golden/cell_d25_b1_L15_trained.npz")
model = uglad_amd.GladParams(1.0, device="cuda:0")
model.load_state_dict({k: torch.from_numpy(np.array(g["param." + k])) for k in ex.PARAM_KEYS})
S = torch.from_numpy(g["S"]).cuda()
theta, loss = uglad_amd.forward_uGLAD(S, model, L=int(g["L"]), INIT_DIAG=0)
loss.backward(); torch.cuda.synchronize()
err = float(np.linalg.norm(theta[0].detach().cpu().numpy() - g["theta_L"][0]) / np.linalg.norm(g["theta_L"][0]))
sd = dict(model.named_parameters())
gerr = max(float(np.linalg.norm(sd[k].grad.cpu().numpy() - g["grad." + k]) / max(np.linalg.norm(g["grad." + k]), 1e-12)) for k in ex.PARAM_KEYS)
one = Collective(); opt = uglad_amd.get_optimizers(model)
def step(train):
    if train:
        opt.zero_grad(); th, ls = um.forward_uGLAD(S, model, L=15, collective=one); ls.backward(); opt.step()
    else:
        with torch.no_grad(): um.forward_uGLAD(S, model, L=15, collective=one)
out = []
for train in (True, False):
    for _ in range(3): step(train)
    torch.cuda.synchronize(); ts = []
    for _ in range(15):
        t = time.perf_counter(); step(train); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    out.append(sorted(ts)[7] * 1e3)
tag = os.path.basename(_lib.get_lib().path) + (" lambda step apart" if os.environ.get("UGLAD_NO_FUSED_LAMBDA") else "")
print(f"{tag:40s} Theta vs golden {err:.2e} worst grad {gerr:.2e}; C1 train {out[0]:.3f} ms/pass, forward-only {out[1]:.3f} ms (medians)")
