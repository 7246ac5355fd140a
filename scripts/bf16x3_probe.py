#!/usr/bin/env python3
"""Experiment record (GPU box): Theta and gradient distance from the reference goldens with a library given by UGLAD_LIB (the
-DUGLAD_BF16X3 development build against the shipped one).  python scripts/bf16x3_probe.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uglad_amd
from uglad_amd.glad.glad_params import PARAM_KEYS

def relF(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

print("library:", uglad_amd._lib.LIB_PATH)
for name in ("cell_d25_b1_L15_trained", "cell_d64_b4_L30_trained", "cell_d96_b2_L30_trained", "cell_d128_b2_L30_trained"):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    m = uglad_amd.GladParams(1.0, device="cuda")
    m.load_state_dict({k: torch.from_numpy(np.array(g["param." + k])) for k in PARAM_KEYS})
    theta, loss = uglad_amd.forward_uGLAD(torch.from_numpy(g["S"]).cuda(), m, L=int(g["L"]), INIT_DIAG=int(g["INIT_DIAG"]))
    loss.backward()
    sd = dict(m.named_parameters())
    worst = max((relF(sd[k].grad.cpu().numpy(), g["grad." + k]), k) for k in PARAM_KEYS)
    print(f"{name}: Theta {relF(theta.detach().cpu().numpy(), g['theta_L']):.2e}  worst gradient {worst[1]} {worst[0]:.2e}")
