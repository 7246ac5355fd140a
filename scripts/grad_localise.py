#!/usr/bin/env python3
"""Diagnostic: where does the gradient error of a trained golden come from?  2x2 hybrid of the kernels (on the host SIMT
emulator, or on the GPU with --gpu) and the fp64 oracle (oracle/glad_exact.py), then stage substitution.

  python scripts/grad_localise.py [golden name] [--gpu]

Rows:  fwd=hip bwd=hip | fwd=hip bwd=f64 | fwd=f64 bwd=hip | fwd=f64 bwd=f64     (all against the reference's golden gradients)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import glad_exact as ex  # noqa: E402
from uglad_amd import _lib  # noqa: E402

GPU = "--gpu" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
name = args[0] if args else "cell_d25_b1_L15_trained"
if GPU:
    lib = _lib.get_lib()
    dev = "cuda"
else:
    import conftest

    lib = conftest.install_emulated_lib()
    dev = "cpu"


def relF(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
L, init_diag = int(g["L"]), int(g["INIT_DIAG"])
p = ex.params64(g, "param.")
pk = torch.tensor(np.concatenate([p[k].ravel() for k in ex.PARAM_KEYS]), dtype=torch.float32, device=dev)
S64 = g["S"].astype(np.float64)
lS64 = g["loss_S"].astype(np.float64) if "loss_S" in g else S64
struct = g["struct"].astype(np.float64) if "struct" in g else None
M, D, _ = S64.shape
MODE = _lib.SQRT_MODES["ns10"]
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
f32 = dict(dtype=torch.float32, device=dev)


def hip_forward():
    S = t(S64)
    Z = torch.empty(L + 1, M, D, D, **f32)
    half, U = torch.empty(L, M, D, D, **f32), torch.empty(L, M, D, D, **f32)
    beta = torch.empty(L, M, D, **f32)
    lam, lam_in = torch.empty(L + 1, **f32), torch.empty(L + 1, 2, **f32)
    nfp, nfs = torch.empty(M, **f32), torch.empty(1, **f32)
    wsp = lib.workspace(M, D, S)
    lib.glad_forward(S, pk, 1.0, init_diag, L, Z, half, U, beta, lam, lam_in, nfp, nfs, wsp, MODE)
    return dict(Z=Z, half=half, U=U, beta=beta, lam=lam, lam_in=lam_in)


def hip_backward(fw, G_L):
    S = t(S64)
    bufs = (torch.empty(M, D, D, **f32), torch.empty(M, D, D, **f32))
    grp, glp, gtp = torch.empty(M, 28, **f32), torch.empty(L, M, **f32), torch.empty(M, **f32)
    grad = torch.empty(42, **f32)
    wsp = lib.workspace(M, D, S)
    lib.glad_backward(t(G_L), S, pk, init_diag, L, fw["Z"], fw["half"], fw["U"], fw["beta"], fw["lam"], fw["lam_in"], bufs[0],
                      bufs[1], grp, glp, gtp, grad, wsp, MODE)
    return unpack(grad.cpu().numpy())


def unpack(v):
    out, o = {}, 0
    for k in ex.PARAM_KEYS:
        n = p[k].size
        out[k] = v[o:o + n].reshape(p[k].shape)
        o += n
    return out


def f64_trace_from(Zs, lams, lam_ins):
    """An oracle trace that replays the backward over a given forward trajectory (Theta_k, lambda_k)."""
    return {"theta_init": Zs[0], "Z_in": [Zs[k] for k in range(L)], "lambdas": list(lams), "lambda_inputs": [tuple(x) for x in lam_ins],
            "theta_L": Zs[L]}


def report(tag, grads):
    e = {k: relF(grads[k], g["grad." + k]) for k in ex.PARAM_KEYS}
    w = max(e, key=e.get)
    print(f"{tag:44s} t_off {e['theta_init_offset']:.2e}  rho0.b {e['rho_l1.0.bias']:.2e}  rho0.w {e['rho_l1.0.weight']:.2e}  "
          f"lam2.w {e['lambda_f.2.weight']:.2e}  worst {w} {e[w]:.2e}")
    return e


print("golden", name, "D", D, "M", M, "L", L, "| grad theta_init_offset (reference):", g["grad.theta_init_offset"])
fw = hip_forward()
ref64, tr64 = ex.glad_forward(S64, p, L, init_diag, loss_S=g["loss_S"] if "loss_S" in g else None, struct=struct, mode="ns10")
Zh = fw["Z"].cpu().numpy().astype(np.float64)
print("Theta_L: hip vs golden %.2e | f64 vs golden %.2e | hip vs f64 %.2e" % (
    max(relF(Zh[L][i], g["theta_L"][i]) for i in range(M)), max(relF(ref64[i], g["theta_L"][i]) for i in range(M)),
    max(relF(Zh[L][i], ref64[i]) for i in range(M))))
print("Theta_0: hip vs f64 %.2e" % relF(Zh[0], tr64["theta_init"]))
for k in (0, 1, L // 2, L - 1):
    print(f"  step {k}: Z_in hip vs f64 {relF(Zh[k], tr64['Z_in'][k]):.2e}  lam {float(fw['lam'][k]):.7f} vs {tr64['lambdas'][k]:.7f}")

# ---- 2 x 2
G_hip = ex.loss_bwd(Zh[L], lS64, struct)  # terminal gradient from the HIP Theta_L in fp64
G_64 = ex.loss_bwd(tr64["theta_L"], lS64, struct)
tr_h = f64_trace_from(Zh, fw["lam"].cpu().numpy().astype(np.float64), fw["lam_in"].cpu().numpy().astype(np.float64))

# the product's own loss backward (uglad_loss_fwd/bwd) for the terminal gradient
thL = fw["Z"][L].contiguous()
lp, tinv = torch.empty(M, **f32), torch.empty(M, D, D, **f32)
wsp = lib.workspace(M, D, thL)
lS = t(lS64)
st = t(struct) if struct is not None else None
lib.loss_fwd(thL, lS, st, lp, tinv, wsp)
Gt = torch.empty(M, D, D, **f32)
lib.loss_bwd(thL, tinv, lS, st, torch.ones(1, **f32), 1.0 / lS.shape[0], Gt)
print("terminal gradient: hip loss_bwd vs f64(loss_bwd at hip Theta_L) %.2e ; f64 at hip Theta vs f64 at f64 Theta %.2e" % (
    relF(Gt.cpu().numpy(), G_hip), relF(G_hip, G_64)))

report("fwd=hip bwd=hip (terminal G hip)", hip_backward(fw, Gt.cpu().numpy()))
report("fwd=hip bwd=hip (terminal G f64@hipTheta)", hip_backward(fw, G_hip))
report("fwd=hip bwd=f64", ex.glad_backward(S64, p, L, tr_h, init_diag, loss_S=lS64 if "loss_S" in g else None, struct=struct, mode="ns10"))
# f64 forward handed to the HIP backward: Z, half, U, beta, lam rounded to fp32
fw64 = {k: v.clone() for k, v in fw.items()}
for k in range(L):
    Zk = tr64["Z_in"][k]
    B = S64 / tr64["lambdas"][k] - Zk
    B = 0.5 * (B + B.transpose(0, 2, 1))
    bb, UU = np.linalg.eigh(B)
    fw64["Z"][k] = t(Zk)
    fw64["half"][k] = t(tr64["theta_half"][k])
    fw64["U"][k] = t(UU)
    fw64["beta"][k] = t(bb)
fw64["Z"][L] = t(tr64["theta_L"])
fw64["lam"] = t(np.array(tr64["lambdas"]))
fw64["lam_in"] = t(np.array(tr64["lambda_inputs"]))
report("fwd=f64 bwd=hip", hip_backward(fw64, G_64))
report("fwd=f64 bwd=f64", ex.glad_backward(S64, p, L, tr64, init_diag, loss_S=lS64 if "loss_S" in g else None, struct=struct, mode="ns10"))

# ---- stage substitution inside the forward: hip trajectory but with individual pieces replaced by fp64
# (a) HIP forward, but U/beta recomputed in fp64 from the HIP Z_in (backward products see an accurate eigenbasis)
fwa = {k: v.clone() for k, v in fw.items()}
for k in range(L):
    lamk = float(fw["lam"][k])
    B = S64 / lamk - Zh[k]
    B = 0.5 * (B + B.transpose(0, 2, 1))
    bb, UU = np.linalg.eigh(B)
    fwa["U"][k] = t(UU)
    fwa["beta"][k] = t(bb)
    fwa["half"][k] = t(ex._theta_half(UU, ex.phi(bb, lamk, "ns10")))
report("fwd=hip Z, U/beta/half f64; bwd=hip", hip_backward(fwa, G_hip))
# (b) Theta_0 in fp64, then forward f64: how much of the error is the initial inverse?
Z0h = Zh[0]
print("Theta_0 error (hip vs f64 inverse): %.2e ; ||Theta_0|| %.3e cond(S+tI) %.3e" % (
    relF(Z0h, tr64["theta_init"]), np.linalg.norm(tr64["theta_init"]), np.linalg.cond(S64[0] + p["theta_init_offset"][0] * np.eye(D))))
# (c) gt alone: -<G0, Theta0^2> with G0 from f64 backward of the hip trajectory, Theta0 hip vs f64

# ---- per-step error anatomy: the HIP cell against the fp64 cell on the SAME input Z_in[k]
print("\nper-step error of the HIP cell given its own input (vs fp64 cell on the same Z_in, lambda):")
for k in (0, 1, 2, L // 2, L - 1):
    lamk = float(fw["lam"][k])
    Zn, hf, UU, bb, nrm = ex.cell_fwd(S64, Zh[k], lamk, p, "ns10")
    hh = fw["half"][k].cpu().numpy().astype(np.float64)
    dH = hh - hf
    C = UU.transpose(0, 2, 1) @ dH @ UU  # error of theta_half in the exact eigenbasis
    Cd = np.array([np.diag(C[m]) for m in range(M)])
    offn = np.sqrt(max(np.linalg.norm(C) ** 2 - np.linalg.norm(Cd) ** 2, 0))
    Uh = fw["U"][k].cpu().numpy().astype(np.float64)
    bh = fw["beta"][k].cpu().numpy().astype(np.float64)
    orth = np.abs(Uh.transpose(0, 2, 1) @ Uh - np.eye(D)).max()
    Bm = S64 / lamk - Zh[k]
    res = np.linalg.norm(Bm @ Uh - Uh * bh[:, None, :]) / np.linalg.norm(Bm)
    print(f"  k={k:2d} half relF {relF(hh, hf):.2e} (eigenbasis: diagonal part {np.linalg.norm(Cd)/np.linalg.norm(hf):.2e}, off-diagonal {offn/np.linalg.norm(hf):.2e})"
          f"  Z_out relF {relF(Zh[k+1], Zn):.2e}  beta err {np.abs(np.sort(bh,axis=1)-bb).max()/np.abs(bb).max():.2e}  orth {orth:.2e}  resid {res:.2e}")
