"""ORACLE (test infrastructure, not product code) -- exact closed-form restatement in numpy fp64.

Only tests/, __graft_entry__.smoke() and bench.py may import this file; `uglad_amd` never does.

The reference evaluates Theta_{k+1/2} = 1/2(-b + (b^T b + 4/lam I)^{1/2}) with a 10-step Newton-Schulz
iteration (uglad/glad/glad.py:139-142, uglad/glad/torch_sqrtm.py:13-29).  Because b = S/lam - Theta is
symmetric, the same quantity is U diag(phi(beta)) U^T with b = U diag(beta) U^T and
phi(beta) = 1/2(-beta + sqrt(beta^2 + 4/lam)).  This file states that exact form, forward AND the
hand-derived backward, one function per kernel of the HIP path so each kernel can
be checked in isolation:

  init_theta / init_theta_bwd      glad.py:103-119   (and d/dt of (S + tI)^-1)
  lambda_nn / lambda_nn_bwd        glad_params.py:51-59,83-95  (inputs are constants: no recurrence, :94)
  cell_fwd / cell_bwd              glad.py:139-147 + glad_params.py:61-81 (rhoNN + soft threshold), Daleckii-Krein
  loss_fwd / loss_bwd              main.py:289-335
  glad_forward / glad_backward     glad.py:74-151 unrolled, returning every saved quantity
  consensus_min                    main.py:673-716

Parity pin: tests/test_oracle_golden.py checks glad_forward/glad_backward against the reference-captured
goldens in tests/golden/ (Theta to <=1e-5 rel-Frobenius, gradients to <=1e-4; the residual is the
reference's own Newton-Schulz truncation error, SURVEY.md section 7 hard part 1).
"""
from __future__ import annotations

import numpy as np

PARAM_KEYS = (
    "theta_init_offset",
    "rho_l1.0.weight", "rho_l1.0.bias", "rho_l1.2.weight", "rho_l1.2.bias", "rho_l1.4.weight", "rho_l1.4.bias",
    "lambda_f.0.weight", "lambda_f.0.bias", "lambda_f.2.weight", "lambda_f.2.bias",
)


def params64(src, prefix: str = "") -> dict:
    return {k: np.asarray(src[prefix + k], dtype=np.float64) for k in PARAM_KEYS}


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


# ----------------------------------------------------------------------------- Theta_0
def init_theta(S, t, INIT_DIAG):
    D = S.shape[-1]
    if INIT_DIAG == 1:
        out = np.zeros_like(S)
        idx = np.arange(D)
        out[:, idx, idx] = 1.0 / (S[:, idx, idx] + t)
        return out
    return np.linalg.inv(S + t * np.eye(D))


def init_theta_bwd(theta0, G0, INIT_DIAG):
    """d loss / d theta_init_offset summed over the batch."""
    if INIT_DIAG == 1:
        d = np.diagonal(theta0, axis1=-2, axis2=-1)
        g = np.diagonal(G0, axis1=-2, axis2=-1)
        return -np.sum(g * d * d)
    return -np.sum(G0 * np.matmul(theta0, theta0).transpose(0, 2, 1))


# ----------------------------------------------------------------------------- LambdaNN
def lambda_nn(p, n, lam_prev):
    x = np.array([n, lam_prev], dtype=np.float64)
    h = np.tanh(p["lambda_f.0.weight"] @ x + p["lambda_f.0.bias"])
    return float(_sig(p["lambda_f.2.weight"] @ h + p["lambda_f.2.bias"])[0])


def lambda_nn_bwd(p, n, lam_prev, g_out, grads):
    x = np.array([n, lam_prev], dtype=np.float64)
    h = np.tanh(p["lambda_f.0.weight"] @ x + p["lambda_f.0.bias"])
    o = _sig(p["lambda_f.2.weight"] @ h + p["lambda_f.2.bias"])
    go = g_out * o * (1 - o)  # (1,)
    grads["lambda_f.2.weight"] += np.outer(go, h)
    grads["lambda_f.2.bias"] += go
    ga = (p["lambda_f.2.weight"].T @ go) * (1 - h * h)
    grads["lambda_f.0.weight"] += np.outer(ga, x)
    grads["lambda_f.0.bias"] += ga


# ----------------------------------------------------------------------------- rhoNN
def _rho_fwd(p, x1, x2, x3):
    W1, b1 = p["rho_l1.0.weight"], p["rho_l1.0.bias"]
    W2, b2 = p["rho_l1.2.weight"], p["rho_l1.2.bias"]
    W3, b3 = p["rho_l1.4.weight"], p["rho_l1.4.bias"]
    X = np.stack(np.broadcast_arrays(x1, x2, x3), axis=-1)
    h1 = np.tanh(X @ W1.T + b1)
    h2 = np.tanh(h1 @ W2.T + b2)
    rho = _sig(h2 @ W3.T + b3)[..., 0]
    return X, h1, h2, rho


def soft_threshold(p, half, S, Z):
    _, _, _, rho = _rho_fwd(p, half, S, Z)
    return np.sign(half) * np.maximum(0.0, np.abs(half) - rho), rho


# ----------------------------------------------------------------------------- cell
NS_ITERS = 10  # torch_sqrtm.py:14,33


def sqrt_spectrum(beta, lam, mode):
    """Eigenvalues r_i of (b^T b + 4/lam I)^{1/2} as the chosen evaluation of the square root sees them.

    mode="exact":  r_i = sqrt(beta_i^2 + 4/lam).
    mode="ns10":   what the reference's 10-step coupled Newton-Schulz iteration (torch_sqrtm.py:13-29)
                   returns.  A = b^T b + cI is a polynomial in the symmetric b, every NS iterate is a polynomial in
                   A, so the iteration acts on each eigenvalue alpha_i = beta_i^2 + c independently:
                   y0 = alpha_i/||A||_F, z0 = 1, 10 x {T = (3 - z y)/2; y = y T; z = T z}, r_i = y10 sqrt(||A||_F),
                   with ||A||_F = sqrt(sum_i alpha_i^2).  O(D) work instead of 30 D^3-flop GEMMs, same numbers.
    """
    c = 4.0 / lam
    alpha = beta * beta + c
    if mode == "exact":
        return np.sqrt(alpha)
    nrm = np.sqrt(np.sum(alpha * alpha, axis=-1, keepdims=True))
    y = alpha / nrm
    z = np.ones_like(y)
    for _ in range(NS_ITERS):
        T = 0.5 * (3.0 - z * y)
        y = y * T
        z = T * z
    return y * np.sqrt(nrm)


def inv_pair_sum(r, mode):
    """K_ij standing for 1/(r_i + r_j), the solution operator of the Lyapunov equation R X + X R = G in R's
    eigenbasis.  mode="exact": exactly that.  mode="ns10": what the reference's 10-step approximate backward
    (torch_sqrtm.py:32-46) computes: with a_i = r_i/||R||_F the iteration
    Q <- (Q(3I - AA) - A^T(A^T Q - Q A))/2, A <- A(3I - AA)/2 multiplies entry (i,j) of U^T Q U by
    (3 - a_i^2 - a_j^2 + a_i a_j)/2 per step, so K_ij = prod_t(...) / (2 ||R||_F)."""
    ri, rj = r[..., :, None], r[..., None, :]
    if mode == "exact":
        return 1.0 / (ri + rj)
    nrm = np.sqrt(np.sum(r * r, axis=-1, keepdims=True))
    a = r / nrm
    P = np.ones(r.shape[:-1] + (r.shape[-1], r.shape[-1]))
    for _ in range(NS_ITERS):
        ai, aj = a[..., :, None], a[..., None, :]
        P = P * (0.5 * (3.0 - ai * ai - aj * aj + ai * aj))
        a = 0.5 * a * (3.0 - a * a)
    return P / (2.0 * nrm[..., None])


def phi(beta, lam, mode="exact"):
    return 0.5 * (-beta + sqrt_spectrum(beta, lam, mode))


# tests/experiments/split_bf16_experiment.py sets this to evaluate the one dense contraction of the cell, (U phi) U^T, in a reduced
# arithmetic (per matrix: hook(U phi, U) -> theta_half); None = fp64.
CONTRACT_HOOK = None


def _theta_half(U, ph):
    A = U * ph[:, None, :]
    if CONTRACT_HOOK is None:
        return A @ U.transpose(0, 2, 1)
    return np.stack([CONTRACT_HOOK(A[m], U[m]) for m in range(U.shape[0])])


def cell_fwd(S, Z, lam, p, mode="exact"):
    """One GLAD cell on a batch.  Returns (Z_next, theta_half, U, beta, per-matrix ||Z_next-theta_half||_F^2)."""
    B = S / lam - Z
    B = 0.5 * (B + B.transpose(0, 2, 1))
    beta, U = np.linalg.eigh(B)
    half = _theta_half(U, phi(beta, lam, mode))
    Zn, _ = soft_threshold(p, half, S, Z)
    nrm = np.sum((Zn - half) ** 2, axis=(1, 2))
    return Zn, half, U, beta, nrm


def divided_differences(beta, lam, mode="exact"):
    """F_ij = (phi(b_i)-phi(b_j))/(b_i-b_j), phi' on the diagonal, in the cancellation-free form
    1/2(-1 + (b_i+b_j) K_ij), K_ij = 1/(r_i+r_j)   [phi(bi)-phi(bj) = 1/2(-(bi-bj) + (ri-rj)) and
    ri-rj = (bi-bj)(bi+bj)/(ri+rj)].  With mode="ns10", K is the reference's approximate Lyapunov operator, which
    is what autograd through theta = 1/2(-b + sqrtm(b^T b + cI)) yields there after symmetrisation:
    G_b = -G/2 + b (G_A + G_A^T), G_A = K o (U^T (G/2) U)."""
    r = sqrt_spectrum(beta, lam, mode)
    bi, bj = beta[..., :, None], beta[..., None, :]
    return 0.5 * (-1.0 + (bi + bj) * inv_pair_sum(r, mode))


def cell_bwd(G_next, S, Z, lam, p, grads, mode="exact"):
    """Backward of one cell.  G_next = dL/dZ_next.  Accumulates the 28 rhoNN gradients into `grads`,
    returns (dL/dZ, dL/dlam summed over the batch)."""
    B = S / lam - Z
    B = 0.5 * (B + B.transpose(0, 2, 1))
    beta, U = np.linalg.eigh(B)
    ph = phi(beta, lam, mode)
    half = _theta_half(U, ph)
    X, h1, h2, rho = _rho_fwd(p, half, S, Z)
    active = (np.abs(half) > rho).astype(np.float64)
    g_rho = -np.sign(half) * active * G_next
    W1, W2, W3 = p["rho_l1.0.weight"], p["rho_l1.2.weight"], p["rho_l1.4.weight"]
    go = (g_rho * rho * (1 - rho))[..., None]  # (...,1)
    grads["rho_l1.4.weight"] += go.reshape(-1, 1).T @ h2.reshape(-1, 3)
    grads["rho_l1.4.bias"] += go.reshape(-1, 1).sum(0)
    ga2 = (go @ W3) * (1 - h2 * h2)
    grads["rho_l1.2.weight"] += ga2.reshape(-1, 3).T @ h1.reshape(-1, 3)
    grads["rho_l1.2.bias"] += ga2.reshape(-1, 3).sum(0)
    ga1 = (ga2 @ W2) * (1 - h1 * h1)
    grads["rho_l1.0.weight"] += ga1.reshape(-1, 3).T @ X.reshape(-1, 3)
    grads["rho_l1.0.bias"] += ga1.reshape(-1, 3).sum(0)
    gx = ga1 @ W1
    G_half = active * G_next + gx[..., 0]
    G_half = 0.5 * (G_half + G_half.transpose(0, 2, 1))
    GZ_direct = gx[..., 2]
    C = U.transpose(0, 2, 1) @ G_half @ U
    F = divided_differences(beta, lam, mode)
    G_B = U @ (C * F) @ U.transpose(0, 2, 1)
    GZ = GZ_direct - G_B
    K = inv_pair_sum(sqrt_spectrum(beta, lam, mode), mode)
    Cd = np.diagonal(C, axis1=-2, axis2=-1)
    Kd = np.diagonal(K, axis1=-2, axis2=-1)  # 1/(2 r_i)
    # d theta_half / d(4/lam) = 1/2 * K_ii per eigenvalue; d(4/lam)/dlam = -4/lam^2
    g_lam = -np.sum(S * G_B) / lam**2 + np.sum(Cd * (-2.0 * Kd / lam**2))
    return GZ, g_lam


# ----------------------------------------------------------------------------- loss
def loss_fwd(theta, S, struct=None):
    Bs = S.shape[0]
    with np.errstate(all="ignore"):
        sign, logdet = np.linalg.slogdet(np.where(np.isfinite(theta), theta, 0.0))
    logdet = np.where(sign > 0, logdet, np.where(sign == 0, -np.inf, np.nan))  # torch.logdet: NaN for det < 0, -inf for det = 0
    logdet = np.where(np.isfinite(theta).all(axis=(1, 2)), logdet, np.nan)
    val = np.sum(-logdet + np.sum(S * theta.transpose(0, 2, 1), axis=(1, 2))) / Bs
    if struct is not None:
        D = theta.shape[-1]
        mask = (1.0 - struct) - np.eye(D)
        val = val + np.sum(np.log(np.cosh(theta * mask))) / Bs
    return float(val)


def _inv_or_nan(theta):
    """Batched inverse; a singular or non-finite matrix gives NaN (torch.logdet's path in the reference yields NaN / -inf there and
    its backward NaN; numpy would raise LinAlgError)."""
    out = np.full_like(theta, np.nan)
    for m in range(theta.shape[0]):
        if np.isfinite(theta[m]).all():
            try:
                out[m] = np.linalg.inv(theta[m])
            except np.linalg.LinAlgError:
                pass
    return out


def loss_bwd(theta, S, struct=None):
    Bs = S.shape[0]
    G = (-_inv_or_nan(theta).transpose(0, 2, 1) + np.broadcast_to(S, theta.shape).transpose(0, 2, 1)) / Bs
    if struct is not None:
        D = theta.shape[-1]
        mask = (1.0 - struct) - np.eye(D)
        G = G + np.tanh(theta * mask) * mask / Bs
    return G


# ----------------------------------------------------------------------------- unrolled
def glad_forward(S, p, L, INIT_DIAG=0, lambda_init=1.0, loss_S=None, struct=None, mode="exact"):
    S = np.asarray(S, dtype=np.float64)
    if S.ndim == 2:
        S = S[None]
    t = float(p["theta_init_offset"][0])
    Z = init_theta(S, t, INIT_DIAG)
    lam = lambda_nn(p, lambda_init, 0.0)
    tr = {"theta_init": Z, "lambdas": [lam], "lambda_inputs": [(lambda_init, 0.0)], "normF": [],
          "theta_half": [], "theta_out": [], "Z_in": []}
    for _ in range(L):
        tr["Z_in"].append(Z)
        Zn, half, U, beta, nrm = cell_fwd(S, Z, lam, p, mode)
        n = float(np.mean(nrm))
        tr["theta_half"].append(half)
        tr["theta_out"].append(Zn)
        tr["normF"].append(n)
        tr["lambda_inputs"].append((n, lam))
        lam = lambda_nn(p, n, lam)
        tr["lambdas"].append(lam)
        Z = Zn
    lS = S if loss_S is None else np.asarray(loss_S, dtype=np.float64)
    tr["loss"] = loss_fwd(Z, lS, struct)
    tr["theta_L"] = Z
    return Z, tr


def glad_backward(S, p, L, tr, INIT_DIAG=0, loss_S=None, struct=None, mode="exact"):
    S = np.asarray(S, dtype=np.float64)
    if S.ndim == 2:
        S = S[None]
    lS = S if loss_S is None else np.asarray(loss_S, dtype=np.float64)
    grads = {k: np.zeros_like(p[k]) for k in PARAM_KEYS}
    G = loss_bwd(tr["theta_L"], lS, struct)
    for k in range(L - 1, -1, -1):
        lam = tr["lambdas"][k]
        G, g_lam = cell_bwd(G, S, tr["Z_in"][k], lam, p, grads, mode)
        n_in, lam_in = tr["lambda_inputs"][k]
        lambda_nn_bwd(p, n_in, lam_in, g_lam, grads)
    grads["theta_init_offset"] += init_theta_bwd(tr["theta_init"], G, INIT_DIAG)
    return grads


def consensus_min(theta_K):
    value = np.min(np.abs(theta_K), axis=0)
    votes = np.sum(np.sign(theta_K), axis=0)
    D = theta_K.shape[-1]
    return (np.where(votes >= 0, 1.0, -1.0) * value).reshape(1, D, D)
