#!/usr/bin/env python3
"""Diagnostic (GPU box): time of Theta_0 = (S + t I)^-1 beyond the eigensolver (uglad_init_theta: ns_ldl_kernel + two Newton steps) per size, and its
accuracy against numpy fp64.  python scripts/ldl_probe.py [D ...]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uglad_amd
from uglad_amd import _lib
from uglad_amd.utils.prepare_data import synthetic_covariance_batch
lib = _lib.get_lib()
pz = np.load(os.path.join(ROOT, "tests", "golden", "params_trained.npz"))
m = uglad_amd.GladParams(1.0, device="cuda"); m.load_state_dict({k: torch.from_numpy(np.array(pz[k])) for k in pz.files})
pk = m.packed().detach().contiguous()
t_off = float(m.state_dict()["theta_init_offset"].item())
for D in [int(a) for a in sys.argv[1:]] or [320, 512, 1024, 2048]:
    Snp = synthetic_covariance_batch(1, D, 4 * D, seed=D)
    S = torch.from_numpy(Snp).cuda(); Z = torch.empty_like(S); wsp = lib.workspace(1, D, S)
    lib.init_theta(S, pk, 0, Z, wsp); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): lib.init_theta(S, pk, 0, Z, wsp)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    ref = np.linalg.inv(Snp[0].astype(np.float64) + t_off * np.eye(D))
    err = np.linalg.norm(Z[0].cpu().numpy() - ref) / np.linalg.norm(ref)
    print(f"D={D:5d}: Theta_0 {dt*1e3:8.2f} ms, rel-Frobenius vs fp64 inverse {err:.2e}", flush=True)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in evs:
        a.record(); lib.init_theta(S, pk, 0, Z, wsp); b.record()
    torch.cuda.synchronize()
    print("         per call, HIP events (ms):", " ".join(f"{a.elapsed_time(b):.2f}" for a, b in evs), flush=True)
