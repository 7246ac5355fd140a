#!/usr/bin/env python3
"""Where do the matrix-iteration path's gradients sit between the reference's fp32 values and the fp64 value of the same function?
Per golden and parameter tensor: |kernel - fp64|, |reference - fp64|, |kernel - reference| (relative Frobenius).
    python tests/experiments/ns_noise_probe.py [golden names]        (GPU box; writes to stdout)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
import uglad_amd  # noqa: E402
from oracle import glad_exact as ex  # noqa: E402

GOLDEN = os.path.join(HERE, "..", "golden")


def relF(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    names = [a for a in sys.argv[1:] if a != "--forced"] or ["cell_d320_b1_L15_trained", "cell_d512_b1_L15_trained"]
    if "--forced" in sys.argv:  # the matrix-iteration path also where the spectral path would run
        from uglad_amd import _lib
        _lib.get_lib().set_matrix_iteration(1)
    for name in names:
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        m = uglad_amd.GladParams(1.0, device="cuda")
        m.load_state_dict({k: torch.from_numpy(np.array(g["param." + k])) for k in ex.PARAM_KEYS})
        S = torch.from_numpy(g["S"]).cuda()
        L = int(g["L"])
        theta, loss = uglad_amd.forward_uGLAD(S, m, L=L, INIT_DIAG=int(g["INIT_DIAG"]))
        loss.backward()
        p = ex.params64(g, "param.")
        t64, tr = ex.glad_forward(g["S"], p, L, int(g["INIT_DIAG"]), mode="ns10")
        g64 = ex.glad_backward(g["S"], p, L, tr, int(g["INIT_DIAG"]), mode="ns10")
        print(f"{name}: Theta kernel-fp64 {relF(theta.detach().cpu().numpy(), t64):.2e}  reference-fp64 {relF(g['theta_L'], t64):.2e}  "
              f"kernel-reference {relF(theta.detach().cpu().numpy(), g['theta_L']):.2e}")
        sd = dict(m.named_parameters())
        for k in ex.PARAM_KEYS:
            got = sd[k].grad.cpu().numpy()
            print(f"   {k:22s} kernel-fp64 {relF(got, g64[k]):.2e}   reference-fp64 {relF(g['grad.' + k], g64[k]):.2e}   "
                  f"kernel-reference {relF(got, g['grad.' + k]):.2e}", flush=True)


if __name__ == "__main__":
    main()
