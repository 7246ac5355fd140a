"""ORACLE (test infrastructure, not product code) -- NS-faithful CPU restatement of uGLAD's hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The shipped package `uglad_amd` never does; its kernels are hand-written HIP.

What it restates, in plain PyTorch-CPU fp32 with the reference's operation sequence
(so it reproduces the reference's *approximation*, not the exact closed form):

  ns_sqrt / _NSSqrt      uglad/glad/torch_sqrtm.py:13-29 (forward, 10 coupled Newton-Schulz steps on
                         A/||A||_F) and :32-46 (the hand-written 10-step approximate Lyapunov backward)
  rho_nn / eta           uglad/glad/glad_params.py:38-49,61-81  (3->3->3->1 tanh/tanh/sigmoid + soft threshold)
  lambda_nn              uglad/glad/glad_params.py:51-59,83-95  (2->3->1; BOTH inputs detached, :94)
  glad                   uglad/glad/glad.py:103-151            (Theta_0 init, L x {b, b^T b + 4/lam I, sqrt, eta, Lambda})
  loss_uGLAD             uglad/main.py:289-335                 (-logdet + trace, divisor = S.shape[0]; log-cosh structure term)
  forward_uGLAD          uglad/main.py:252-286
  consensus_min          uglad/main.py:673-716 (type="min")

The only liberty taken: the reference loops over the batch in Python calling the sqrt one matrix at a
time (glad.py:53-57); here the same per-matrix arithmetic runs as batched `bmm`, which makes this a
*stronger* CPU baseline than the reference itself.

Parity pin: tests/test_oracle_golden.py checks every function here against the vectors in
tests/golden/*.npz, which tests/golden/make_goldens.py captured from the real reference
(torch 2.10.0 CPU) in the build container.
"""
from __future__ import annotations

import torch

NS_ITERS = 10  # torch_sqrtm.py:14,33

PARAM_KEYS = (
    "theta_init_offset",
    "rho_l1.0.weight", "rho_l1.0.bias", "rho_l1.2.weight", "rho_l1.2.bias", "rho_l1.4.weight", "rho_l1.4.bias",
    "lambda_f.0.weight", "lambda_f.0.bias", "lambda_f.2.weight", "lambda_f.2.bias",
)


class _NSSqrt(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A):  # A: (B, D, D)
        D = A.shape[-1]
        nrm = torch.linalg.matrix_norm(A).reshape(-1, 1, 1)
        Y = A / nrm
        eye = torch.eye(D, dtype=A.dtype).expand_as(A)
        Z = eye.clone()
        for _ in range(NS_ITERS):
            T = 0.5 * (3.0 * eye - torch.bmm(Z, Y))
            Y = torch.bmm(Y, T)
            Z = torch.bmm(T, Z)
        R = Y * torch.sqrt(nrm)
        ctx.save_for_backward(R)
        return R

    @staticmethod
    def backward(ctx, G):
        (R,) = ctx.saved_tensors
        D = R.shape[-1]
        nrm = torch.linalg.matrix_norm(R).reshape(-1, 1, 1)
        A = R / nrm
        Q = G / nrm
        eye = torch.eye(D, dtype=R.dtype).expand_as(R)
        for _ in range(NS_ITERS):
            AA = torch.bmm(A, A)
            At = A.transpose(-1, -2)
            Q = 0.5 * (torch.bmm(Q, 3.0 * eye - AA) - torch.bmm(At, torch.bmm(At, Q) - torch.bmm(Q, A)))
            A = 0.5 * torch.bmm(A, 3.0 * eye - AA)
        return 0.5 * Q


def ns_sqrt(A: torch.Tensor) -> torch.Tensor:
    if A.dim() == 2:
        return _NSSqrt.apply(A[None])[0]
    return _NSSqrt.apply(A)


def rho_nn(p: dict, x: torch.Tensor) -> torch.Tensor:
    """x (..., 3) -> rho (..., 1)"""
    h = torch.tanh(x @ p["rho_l1.0.weight"].T + p["rho_l1.0.bias"])
    h = torch.tanh(h @ p["rho_l1.2.weight"].T + p["rho_l1.2.bias"])
    return torch.sigmoid(h @ p["rho_l1.4.weight"].T + p["rho_l1.4.bias"])


def eta(p: dict, X: torch.Tensor, S: torch.Tensor, Zprev: torch.Tensor) -> torch.Tensor:
    Sb = S.expand_as(X)
    feat = torch.stack((X, Sb, Zprev), dim=-1)
    rho = rho_nn(p, feat)[..., 0]
    return torch.sign(X) * torch.clamp_min(torch.abs(X) - rho, 0.0)


def lambda_nn(p: dict, normF, prev_lambda) -> torch.Tensor:
    x = torch.tensor([float(normF), float(prev_lambda)], dtype=p["lambda_f.0.weight"].dtype)  # detached on purpose
    h = torch.tanh(p["lambda_f.0.weight"] @ x + p["lambda_f.0.bias"])
    return torch.sigmoid(p["lambda_f.2.weight"] @ h + p["lambda_f.2.bias"])


def _inverse(A: torch.Tensor) -> torch.Tensor:
    """(S + tI)^-1 of glad.py:115.  The reference calls torch.inverse (fp32 LU).  Here the value comes from numpy's fp64 LAPACK,
    rounded to fp32, and one Newton step X(2I - AX) re-attaches it to autograd (value X, derivative -X dA X: exactly those of
    the inverse).  Reason: this oracle also runs on the GPU node's host, where torch's CPU LU was observed to fail at D=256
    ("Pivots given to lu_solve ...", NaN logdet); an exact primitive must not depend on the host's LAPACK build."""
    import numpy as np

    X = torch.from_numpy(np.linalg.inv(A.detach().numpy().astype(np.float64))).to(A.dtype)  # (fp32 as the reference; fp64 for forensics)
    eye = torch.eye(A.shape[-1], dtype=A.dtype).expand_as(A)
    return torch.bmm(X, 2.0 * eye - torch.bmm(A, X))


class _LogDet(torch.autograd.Function):
    """torch.logdet semantics (NaN for det < 0, -inf for det = 0) with the value and the gradient Theta^-T from numpy fp64."""

    @staticmethod
    def forward(ctx, theta):
        import numpy as np

        a = theta.detach().numpy().astype(np.float64)
        sign, lad = np.linalg.slogdet(a)
        out = np.where(sign > 0, lad, np.where(sign == 0, -np.inf, np.nan))
        ok = np.isfinite(out)
        inv = np.zeros_like(a)
        if ok.any():
            inv[ok] = np.linalg.inv(a[ok])
        inv[~ok] = np.nan
        ctx.save_for_backward(torch.from_numpy(inv.transpose(0, 2, 1).copy()).to(theta.dtype))
        return torch.from_numpy(out).to(theta.dtype)

    @staticmethod
    def backward(ctx, g):
        (inv_t,) = ctx.saved_tensors
        return g.reshape(-1, 1, 1) * inv_t


def glad(S: torch.Tensor, p: dict, lambda_init: float = 1.0, L: int = 15, INIT_DIAG: int = 0, trace: dict | None = None):
    if S.dim() == 2:
        S = S[None]
    D = S.shape[-1]
    eye = torch.eye(D, dtype=S.dtype).expand_as(S)
    t = p["theta_init_offset"]
    if INIT_DIAG == 1:
        theta = torch.diag_embed(1.0 / (torch.diagonal(S, dim1=-2, dim2=-1) + t))
    else:
        theta = _inverse(S + t * eye)
    lam = lambda_nn(p, lambda_init, 0.0)
    if trace is not None:
        trace.update(theta_init=theta.detach().clone(), lambdas=[float(lam.detach())], normF=[], theta_half=[], theta_out=[])
    for _ in range(L):
        b = S / lam - theta
        A = torch.bmm(b.transpose(-1, -2), b) + (4.0 / lam) * eye
        half = 0.5 * (ns_sqrt(A) - b)
        new = eta(p, half, S, theta)
        nF = torch.mean(torch.sum((new - half) ** 2, dim=(1, 2))).item()
        theta = new
        lam = lambda_nn(p, nF, float(lam.detach()))
        if trace is not None:
            trace["lambdas"].append(float(lam.detach()))
            trace["normF"].append(nF)
            trace["theta_half"].append(half.detach().clone())
            trace["theta_out"].append(new.detach().clone())
    return theta


def loss_uGLAD(theta: torch.Tensor, S: torch.Tensor, struct_theta: torch.Tensor | None = None) -> torch.Tensor:
    B, D, _ = S.shape
    t1 = -_LogDet.apply(theta)
    t2 = torch.sum(S * theta.transpose(-1, -2), dim=(1, 2))
    loss = torch.sum(t1 + t2) / B
    if struct_theta is not None:
        mask = (1.0 - struct_theta) - torch.eye(D, dtype=theta.dtype).expand(B, -1, -1)
        loss = loss + torch.sum(torch.log(torch.cosh(theta * mask))) / B
    return loss


def forward_uGLAD(S, p, L=15, INIT_DIAG=0, loss_Sb=None, struct_theta=None, trace=None):
    theta = glad(S, p, L=L, INIT_DIAG=INIT_DIAG, trace=trace)
    loss = loss_uGLAD(theta, S if loss_Sb is None else loss_Sb, struct_theta)
    return theta, loss


def consensus_min(theta_K: torch.Tensor) -> torch.Tensor:
    value = torch.min(torch.abs(theta_K), 0)[0]
    votes = torch.sum(torch.sign(theta_K), 0)
    sign = torch.where(votes >= 0, torch.ones_like(votes), -torch.ones_like(votes))
    D = theta_K.shape[-1]
    return (sign * value).reshape(1, D, D)


def params_from_npz(npz, prefix: str = "", requires_grad: bool = False) -> dict:
    out = {}
    for k in PARAM_KEYS:
        t = torch.tensor(npz[prefix + k], dtype=torch.float32)
        t.requires_grad_(requires_grad)
        out[k] = t
    return out
