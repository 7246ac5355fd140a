#!/usr/bin/env python3
"""Diagnostic (GPU box): the matrix-iteration path at large D -- one matrix, L = 2, forward + loss + backward: time per pass and the distance of
Theta_L / the 42 gradients from the fp64 oracle (sizes where the oracle finishes in seconds).  python scripts/ns_large_probe.py [D ...]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uglad_amd
from oracle import glad_exact as ex  # (the checker)
from uglad_amd import _lib
from uglad_amd.utils.prepare_data import synthetic_covariance_batch
def relF(a, b): return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30))
g = np.load(os.path.join(ROOT, "tests", "golden", "params_trained.npz"))
p64 = ex.params64(g, "")
print("library:", _lib.get_lib().path, "max D", _lib.get_lib().max_dim)
for D in [int(a) for a in sys.argv[1:]] or [512, 1024, 1100, 1536, 2048]:
    Snp = synthetic_covariance_batch(1, D, 4 * D, seed=D)
    S = torch.from_numpy(Snp).cuda()
    def step():
        m = uglad_amd.GladParams(1.0, device="cuda")
        m.load_state_dict({k: torch.from_numpy(np.array(g[k])) for k in ex.PARAM_KEYS})
        th, ls = uglad_amd.forward_uGLAD(S, m, L=2)
        ls.backward()
        return th, ls, m
    th, ls, m = step(); torch.cuda.synchronize()
    t = time.perf_counter(); th, ls, m = step(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    line = f"D={D:5d}: {dt*1e3:8.1f} ms per training pass (L = 2), loss {ls.item():.6g}"
    if D <= 1600:
        t0 = time.perf_counter()
        ref, tr = ex.glad_forward(Snp, p64, 2, 0, mode="ns10")
        grads = ex.glad_backward(Snp, p64, 2, tr, 0, mode="ns10")
        sd = dict(m.named_parameters())
        worst = max(relF(sd[k].grad.cpu().numpy(), grads[k]) for k in ex.PARAM_KEYS)
        line += f"; Theta vs fp64 oracle {relF(th[0].detach().cpu().numpy(), ref[0]):.2e}, loss rel {abs(ls.item()-tr['loss'])/abs(tr['loss']):.1e}, worst gradient {worst:.2e} (oracle {time.perf_counter()-t0:.0f} s)"
    print(line, flush=True)
