#!/usr/bin/env python3
"""Diagnostic (GPU box): ADVICE round 2 -- tests/test_gpu_parity.py::test_many_workgroups_per_matrix_ragged_sizes was written with the
default synthetic inputs (N = 500 samples) and changed to N = 2048 without a record of what failed.  This re-runs the N = 500 case for
every (D, M) of the test and prints, per kernel shape, what each assertion of the test sees, next to the conditioning of the input."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uglad_amd  # noqa: E402
from oracle import glad_exact as ex  # noqa: E402  (diagnostic script, not product)
from uglad_amd import _lib  # noqa: E402
from uglad_amd.utils.prepare_data import synthetic_covariance_batch  # noqa: E402

lib = _lib.get_lib()
g = np.load(os.path.join(ROOT, "tests", "golden", "params_trained.npz"))


def model():
    m = uglad_amd.GladParams(1.0, device="cuda")
    m.load_state_dict({k: torch.from_numpy(np.array(g[k])) for k in ex.PARAM_KEYS})
    return m


def relF(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


t_off = float(g["theta_init_offset"][0])
for N in (500, 2048):
    for D, M in [(130, 3), (160, 1), (161, 2), (192, 1), (193, 3), (224, 2), (255, 1)]:
        Sn = synthetic_covariance_batch(M, D, N, seed=1000 + D)
        w = np.linalg.eigvalsh(Sn.astype(np.float64))
        S = torch.from_numpy(Sn).cuda()
        out = {}
        for wide in (0, 1):
            lib.set_wide_mode(wide)
            try:
                m = model()
                theta, loss = uglad_amd.forward_uGLAD(S, m, L=6)
                loss.backward()
            finally:
                lib.set_wide_mode(-1)
            out[wide] = (theta.detach().cpu().numpy(), torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy(), loss.item())
        p64 = {k: np.asarray(g[k], np.float64) for k in ex.PARAM_KEYS}
        ref, tr = ex.glad_forward(Sn, p64, 6, 0, mode="ns10")
        print(f"N={N} D={D} M={M}: min eig(S) {w.min():.3e} (t = {t_off:.2e}: min eig(S+tI) {w.min()+t_off:.3e}) cond(S+tI) {(w.max()+t_off)/(w.min()+t_off):.2e} | "
              f"finite theta {np.isfinite(out[0][0]).all()}/{np.isfinite(out[1][0]).all()} loss {out[0][2]:.6g}/{out[1][2]:.6g} (f64 oracle {tr['loss']:.6g}) | "
              f"theta wide vs one-wg {max(relF(out[1][0][i], out[0][0][i]) for i in range(M)):.2e}; one-wg vs f64 {max(relF(out[0][0][i], ref[i]) for i in range(M)):.2e}; "
              f"wide vs f64 {max(relF(out[1][0][i], ref[i]) for i in range(M)):.2e} | grads wide vs one-wg {relF(out[1][1], out[0][1]):.2e}", flush=True)
