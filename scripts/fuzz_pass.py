#!/usr/bin/env python3
"""Diagnostic (GPU box): random (M, D, L, INIT_DIAG, parameters, input kind) -- the whole unrolled pass against the fp64 oracle of the same function.
Parameters are the trained or the fresh set, randomly perturbed (so thresholds, lambda and the offset t move); inputs are covariances of few samples
(N from D/2 to 4 D, repaired like the reference does), correlation-like matrices, scaled ones.  Flags: Theta_L > 3e-5 or gradients > 2e-3 relative
Frobenius where the oracle is finite and well away from a singular Theta_L; finite here where the oracle is not (or the other way round).
    python scripts/fuzz_pass.py [seed=0] [cases=150] [maxD=64] [minD=1] [--extras]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uglad_amd
from oracle import glad_exact as ex

def relF(a, b): return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / max(np.linalg.norm(np.asarray(b, np.float64)), 1e-300))
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
maxD = int(sys.argv[3]) if len(sys.argv) > 3 else 64
minD = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4].isdigit() else 1
rng = np.random.default_rng(seed)
sets = {k: np.load(os.path.join(ROOT, "tests", "golden", f"params_{k}.npz")) for k in ("trained", "fresh")}
bad = 0; worst_t = worst_g = 0.0; nonfinite = 0
for case in range(cases):
    D = int(rng.integers(minD, maxD + 1)); M = int(rng.integers(1, 5 if maxD <= 256 else 3)); L = int(rng.integers(1, 9)); diag = int(rng.integers(0, 2))
    which = "trained" if rng.random() < 0.6 else "fresh"
    pert = float(rng.choice([0.0, 0.02, 0.1]))
    sd = {k: np.array(sets[which][k], np.float32) for k in sets[which].files}
    sd = {k: (v * (1.0 + pert * rng.standard_normal(v.shape))).astype(np.float32) for k, v in sd.items()}
    kind = str(rng.choice(["few samples", "many samples", "correlation", "scaled x8", "scaled /8"]))
    S = np.empty((M, D, D), np.float64)
    for m in range(M):
        N = max(2, int(D * (0.5 if kind == "few samples" else 4.0)))
        X = rng.standard_normal((N, D)) @ (np.eye(D) + 0.3 * rng.standard_normal((D, D)) / np.sqrt(D))
        C = np.cov(X, rowvar=False, bias=True).reshape(D, D)
        if kind == "correlation":
            s = np.sqrt(np.clip(np.diag(C), 1e-12, None)); C = C / s[:, None] / s[None, :]
        w = np.linalg.eigvalsh(C)
        if w.min() <= 1e-6: C = C + (0.1 - w.min()) * np.eye(D)  # (the reference's repair, prepare_data.py)
        if kind == "scaled x8": C = 8.0 * C
        if kind == "scaled /8": C = C / 8.0
        S[m] = C
    S32 = np.ascontiguousarray(S.astype(np.float32))
    model = uglad_amd.GladParams(1.0, device="cuda")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    p64 = ex.params64(sd)
    # the loss against another covariance (the missing-data mode: one full-data matrix for the whole batch, main.py:303-316) and / or with the
    # structure penalty (a symmetric 0/1 prior, main.py:288-301)
    extra = str(rng.choice(["plain", "plain", "loss_S one", "struct", "both"])) if "--extras" in sys.argv else "plain"
    kw, okw = {}, {}
    if extra in ("loss_S one", "both"):
        ls = np.ascontiguousarray((S32.mean(axis=0, keepdims=True) + 0.05 * np.eye(D, dtype=np.float32)[None]).astype(np.float32))
        kw["loss_Sb"] = torch.from_numpy(ls).cuda(); okw["loss_S"] = ls.astype(np.float64)
    if extra in ("struct", "both"):
        st = (rng.random((D, D)) < 0.3).astype(np.float32); st = np.maximum(st, st.T); np.fill_diagonal(st, 1.0)
        st = np.ascontiguousarray(st[None])
        kw["struct_theta"] = torch.from_numpy(st).cuda(); okw["struct"] = st.astype(np.float64)
    theta, loss = uglad_amd.forward_uGLAD(torch.from_numpy(S32).cuda(), model, L=L, INIT_DIAG=diag, **kw)
    loss.backward(); torch.cuda.synchronize()
    th64, tr = ex.glad_forward(S32.astype(np.float64), p64, L, diag, mode="ns10", **okw)
    l64 = float(ex.loss_fwd(th64, okw.get("loss_S", S32.astype(np.float64)), okw.get("struct")))
    tag = f"case {case}: M={M} D={D} L={L} diag={diag} {which} pert {pert} {kind} {extra}"
    if not (np.isfinite(th64).all() and np.isfinite(l64)):
        nonfinite += 1
        if np.isfinite(loss.item()):
            bad += 1; print(tag, f"oracle loss {l64} but kernels {loss.item()}   <--", flush=True)
        continue
    wmin = np.linalg.eigvalsh(th64).min(axis=1).min() / np.abs(th64).max()
    g64 = ex.glad_backward(S32.astype(np.float64), p64, L, tr, diag, mode="ns10", **okw)
    spar = dict(model.named_parameters())
    got = np.concatenate([spar[k].grad.cpu().numpy().astype(np.float64).reshape(-1) for k in ex.PARAM_KEYS])
    ref = np.concatenate([np.asarray(g64[k], np.float64).reshape(-1) for k in ex.PARAM_KEYS])
    et, eg = relF(theta.detach().cpu().numpy(), th64), relF(got, ref)
    near_singular = wmin < 1e-4  # (logdet's gradient Theta^-1 blows up: not a meaningful comparison)
    if not near_singular:
        worst_t, worst_g = max(worst_t, et), max(worst_g, eg)
    flag = (not np.isfinite(loss.item())) or et > 3e-5 or (eg > 2e-3 and not near_singular)
    if flag:
        bad += 1; print(tag, f"Theta {et:.2e} gradients {eg:.2e} loss {loss.item():.6g} (fp64 {l64:.6g}) min eig/max {wmin:.1e}   <--", flush=True)
print(f"seed {seed}: {cases} cases, D = {minD} ... {maxD}, {nonfinite} with a non-finite oracle loss; worst Theta {worst_t:.2e}, worst gradients {worst_g:.2e}; flagged {bad}")
