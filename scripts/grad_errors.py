#!/usr/bin/env python3
"""Diagnostic (GPU box): gradient errors of every cell golden (D <= 128 unless --all) against the reference's outputs, for the library
UGLAD_LIB points at (default: the shipped one).  Used to compare build variants (scripts/dev_build.sh)."""
import glob, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uglad_amd
from uglad_amd import _lib
KEYS = ["theta_init_offset", "rho_l1.0.weight", "rho_l1.0.bias", "rho_l1.2.weight", "rho_l1.2.bias", "rho_l1.4.weight", "rho_l1.4.bias",
        "lambda_f.0.weight", "lambda_f.0.bias", "lambda_f.2.weight", "lambda_f.2.bias"]
def relF(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
print("library:", _lib.get_lib().path)
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "cell_*.npz"))):
    g = np.load(path); name = os.path.basename(path)[:-4]
    if g["S"].shape[-1] > _lib.get_lib().max_dim or (g["S"].shape[-1] > 128 and "--all" not in sys.argv):
        continue
    m = uglad_amd.GladParams(1.0, device="cuda")
    m.load_state_dict({k: torch.from_numpy(np.array(g["param." + k])) for k in KEYS})
    kw = {}
    if "loss_S" in g: kw["loss_Sb"] = torch.from_numpy(g["loss_S"]).cuda()
    if "struct" in g: kw["struct_theta"] = torch.from_numpy(g["struct"]).cuda()
    th, ls = uglad_amd.forward_uGLAD(torch.from_numpy(g["S"]).cuda(), m, L=int(g["L"]), INIT_DIAG=int(g["INIT_DIAG"]), **kw)
    ls.backward()
    sd = dict(m.named_parameters())
    e = {k: relF(sd[k].grad.cpu().numpy(), g["grad." + k]) for k in KEYS}
    w = max(e, key=e.get)
    te = max(relF(th[i].detach().cpu().numpy(), g["theta_L"][i]) for i in range(th.shape[0]))
    print(f"{name:34s} Theta {te:.2e}  theta_init_offset {e['theta_init_offset']:.2e}  rho_l1.0.bias {e['rho_l1.0.bias']:.2e}  worst {w} {e[w]:.2e}")
