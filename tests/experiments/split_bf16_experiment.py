#!/usr/bin/env python3
"""What would a split-bf16 (bf16 x 3: hi*hi + hi*lo + lo*hi, fp32 accumulate) contraction cost in accuracy on THIS path?

BASELINE.json's config 3 line says "bf16 MFMA back-transform"; SURVEY.md section 7 (hard part 3) estimated 2-3e-6 for the split
form.  Here it is measured on the reference-captured goldens: the fp64 spectral oracle (oracle/glad_exact.py, mode ns10)
runs the whole unrolled pass with ONLY the contraction theta_half = (U phi) U^T evaluated the way a bf16 x 3 MFMA kernel
would -- operands rounded to fp32, split into bf16 hi + bf16 lo, three products accumulated in fp32 -- and, for
comparison, with the operands rounded to fp32 and an fp32 product (what the f32 MFMA path does) and with plain bf16
operands.  Everything else stays fp64, so the numbers isolate the contraction.  CPU only (numpy); prints a table.

    python tests/experiments/split_bf16_experiment.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import glad_exact as ex  # noqa: E402  (test infrastructure using test infrastructure; nothing shipped imports this)


def bf16(x):
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def contract(A, B, mode):
    """A @ B^T the way the chosen arithmetic would do it (fp32 accumulate emulated by an fp64 sum of fp32-exact products)."""
    A32, B32 = A.astype(np.float32), B.astype(np.float32)
    if mode == "f32":
        return (A32.astype(np.float64) @ B32.astype(np.float64).T).astype(np.float32).astype(np.float64)
    Ah, Bh = bf16(A32), bf16(B32)
    if mode == "bf16":
        return (Ah.astype(np.float64) @ Bh.astype(np.float64).T).astype(np.float32).astype(np.float64)
    Al, Bl = bf16(A32 - Ah), bf16(B32 - Bh)
    f = lambda X, Y: X.astype(np.float64) @ Y.astype(np.float64).T  # noqa: E731
    return (f(Ah, Bh) + f(Ah, Bl) + f(Al, Bh)).astype(np.float32).astype(np.float64)


def run(name, mode):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    p = ex.params64(g, "param.")
    # the oracle forms theta_half in one place (glad_exact._theta_half) and offers a hook there
    ex.CONTRACT_HOOK = (lambda Uphi, U: contract(Uphi, U, mode)) if mode != "f64" else None
    try:
        theta, tr = ex.glad_forward(g["S"], p, int(g["L"]), int(g["INIT_DIAG"]), mode="ns10")
        grads = ex.glad_backward(g["S"], p, int(g["L"]), tr, int(g["INIT_DIAG"]), mode="ns10")
    finally:
        ex.CONTRACT_HOOK = None
    relF = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30))  # noqa: E731
    th = max(relF(theta[i], g["theta_L"][i]) for i in range(theta.shape[0]))
    gr = max(relF(grads[k], g["grad." + k]) for k in ex.PARAM_KEYS)
    return th, gr


if __name__ == "__main__":
    print(f"{'golden':30s} {'contraction':>12s} {'Theta_L relF vs reference':>26s} {'worst gradient relF':>20s}")
    for name in ("cell_d25_b1_L15_trained", "cell_d64_b4_L30_trained", "cell_d128_b2_L30_trained"):
        for mode in ("f64", "f32", "bf16x3", "bf16"):
            th, gr = run(name, mode)
            print(f"{name:30s} {mode:>12s} {th:26.2e} {gr:20.2e}", flush=True)


# ---- round 3: the same question for the SHIFTED form theta_half = -alpha b + (U psi) U^T (uglad_amd/csrc/glad_device.h, shifted_spectrum):
# only the small remainder U psi U^T goes through the contraction, so its rounding error is scaled down by ||psi|| / ||phi||.
def run_shifted(name, mode):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    p = ex.params64(g, "param.")
    orig = ex.cell_fwd

    def cell_fwd(S, Z, lam, p_, sqrt_mode="exact"):
        B = S / lam - Z
        B = 0.5 * (B + B.transpose(0, 2, 1))
        beta, U = np.linalg.eigh(B)
        ph = ex.phi(beta, lam, sqrt_mode)
        alpha = np.clip(-(ph * beta).sum(-1) / (beta * beta).sum(-1), 0.0, 1.0)[:, None]
        psi = ph + alpha * beta
        rem = np.stack([contract(U[m] * psi[m][None, :], U[m], mode) if mode != "f64" else (U[m] * psi[m][None, :]) @ U[m].T
                        for m in range(U.shape[0])])
        half = rem - alpha[:, :, None] * B
        Zn, _ = ex.soft_threshold(p_, half, S, Z)
        return Zn, half, U, beta, np.sum((Zn - half) ** 2, axis=(1, 2))

    ex.cell_fwd = cell_fwd
    try:
        theta, tr = ex.glad_forward(g["S"], p, int(g["L"]), int(g["INIT_DIAG"]), mode="ns10")
    finally:
        ex.cell_fwd = orig
    grads = ex.glad_backward(g["S"], p, int(g["L"]), tr, int(g["INIT_DIAG"]), mode="ns10")
    relF = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30))  # noqa: E731
    th = max(relF(theta[i], g["theta_L"][i]) for i in range(theta.shape[0]))
    gr = max(relF(grads[k], g["grad." + k]) for k in ex.PARAM_KEYS)
    return th, gr


if __name__ == "__main__":
    print("\nshifted form (round 3): only U psi U^T goes through the contraction")
    for name in ("cell_d25_b1_L15_trained", "cell_d64_b4_L30_trained", "cell_d128_b2_L30_trained", "cell_d64_b4_L30_fresh", "cell_d128_b2_L30_fresh"):
        for mode in ("f64", "f32", "bf16x3", "bf16"):
            th, gr = run_shifted(name, mode)
            print(f"{name:30s} {mode:>12s} {th:26.2e} {gr:20.2e}", flush=True)
