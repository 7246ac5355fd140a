"""Pin the oracle (oracle/glad_ns.py, oracle/glad_exact.py) to vectors captured from the real reference.

CPU-only.  Tolerances: the NS-faithful restatement repeats the reference's arithmetic, so it must agree to
fp32 round-off (<=2e-6 rel-Frobenius, <=1e-5 on gradients, the batched-vs-per-matrix mm order being the
only difference).  The fp64 spectral restatement in mode="ns10" applies the reference's
Newton-Schulz truncation per eigenvalue and must agree everywhere (<=3e-5 on Theta, <=1e-4 on gradients); in
mode="exact" it agrees only where the reference's NS-10 has converged (fresh parameters).
"""
import glob
import json
import os

import numpy as np
import pytest
import torch

from oracle import glad_exact as ex
from oracle import glad_ns as ns

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CELLS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "cell_*.npz")))
SMALL = [c for c in CELLS if "d128" not in c and "d256" not in c]


def relF(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def max_relF_batch(a, b):
    return max(relF(a[i], b[i]) for i in range(b.shape[0]))


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def test_goldens_present():
    assert len(CELLS) >= 10
    for f in ("params_fresh", "params_trained", "consensus", "fit_direct_d25", "fit_cv_d16",
              "fit_missing_d20", "fit_multitask_d20_k3"):
        assert os.path.exists(os.path.join(GOLDEN, f + ".npz"))


@pytest.mark.parametrize("name", CELLS)
def test_ns_restatement_matches_reference(name):
    g = load(name)
    if "d256" in name or "d128" in name:
        torch.set_num_threads(8)
    p = ns.params_from_npz(g, "param.", requires_grad=True)
    S = torch.from_numpy(g["S"])
    kw = {}
    if "loss_S" in g:
        kw["loss_Sb"] = torch.from_numpy(g["loss_S"])
    if "struct" in g:
        kw["struct_theta"] = torch.from_numpy(g["struct"])
    tr = {}
    theta, loss = ns.forward_uGLAD(S, p, L=int(g["L"]), INIT_DIAG=int(g["INIT_DIAG"]), trace=tr, **kw)
    loss.backward()
    # two fp32 evaluations of the matrix iteration (this one batched, other BLAS calls) drift apart with D: 8.5e-6 at D = 512, where
    # each is 1e-5 from the fp64 value (tests/golden/grad_noise_floor.json); gradients likewise, bounded by that file's floor
    big = S.shape[-1] > 256
    floor = json.load(open(os.path.join(GOLDEN, "grad_noise_floor.json"))).get(name, {"grads": {"": 0.0}})
    gtol = max(5e-5, 2.0 * max(floor["grads"].values())) if big else 5e-5
    assert max_relF_batch(theta.detach().numpy(), g["theta_L"]) < (2e-5 if big else 5e-6)
    assert abs(loss.item() - float(g["loss"])) < 2e-5 * max(1.0, abs(float(g["loss"])))
    np.testing.assert_allclose(np.array(tr["lambdas"]), g["lambdas"], rtol=2e-6, atol=1e-7)
    for k in g["keep_k"]:
        assert max_relF_batch(tr["theta_half"][k].numpy(), g[f"theta_half_{k}"]) < 5e-6
        assert max_relF_batch(tr["theta_out"][k].numpy(), g[f"theta_out_{k}"]) < 5e-6
    for key in ns.PARAM_KEYS:
        ref = g["grad." + key]
        got = p[key].grad.numpy()
        assert relF(got, ref) < gtol or np.abs(got - ref).max() < 1e-6, key


@pytest.mark.parametrize("name", CELLS)
def test_spectral_ns10_matches_reference_everywhere(name):
    """mode="ns10" (the reference's Newton-Schulz truncation applied per eigenvalue) is the parity target of the HIP
    path: it must track the reference in every regime, converged or not."""
    g = load(name)
    p = ex.params64(g, "param.")
    L, diag = int(g["L"]), int(g["INIT_DIAG"])
    loss_S = g["loss_S"] if "loss_S" in g else None
    struct = g["struct"] if "struct" in g else None
    theta, tr = ex.glad_forward(g["S"], p, L, diag, loss_S=loss_S, struct=struct, mode="ns10")
    assert max_relF_batch(theta, g["theta_L"]) < 3e-5  # measured <=1.9e-5 (D=256: the reference's own fp32 round-off)
    assert abs(tr["loss"] - float(g["loss"])) < 2e-5 * max(1.0, abs(float(g["loss"])))
    np.testing.assert_allclose(np.array(tr["lambdas"]), g["lambdas"], rtol=5e-6, atol=1e-7)
    for k in g["keep_k"]:
        assert max_relF_batch(tr["theta_half"][k], g[f"theta_half_{k}"]) < 1e-5
        assert max_relF_batch(tr["theta_out"][k], g[f"theta_out_{k}"]) < 1e-5
    if g["S"].shape[-1] > 128:
        return
    grads = ex.glad_backward(g["S"], p, L, tr, diag, loss_S=loss_S, struct=struct, mode="ns10")
    for key in ex.PARAM_KEYS:
        ref = g["grad." + key]
        assert relF(grads[key], ref) < 1e-4 or np.abs(grads[key] - ref).max() < 2e-6, (key, grads[key], ref)


@pytest.mark.parametrize("name", [c for c in CELLS if "fresh" in c and "d256" not in c and "d288" not in c and "d1024" not in c])  # (d288, d1024: ten steps fall 1.1e-4 ... short)
def test_exact_closed_form_matches_reference_when_ns_converged(name):
    """With fresh parameters (lambda ~ 0.34, cond(b^T b + 4/lam I) small) the reference's NS-10 has converged, so the
    exact closed form agrees too; with trained parameters it does not (measured: Theta 2.3e-3, gradients 68 % off at
    D=128) -- which is why ns10 is the default of the HIP path."""
    g = load(name)
    p = ex.params64(g, "param.")
    L, diag = int(g["L"]), int(g["INIT_DIAG"])
    loss_S = g["loss_S"] if "loss_S" in g else None
    struct = g["struct"] if "struct" in g else None
    theta, tr = ex.glad_forward(g["S"], p, L, diag, loss_S=loss_S, struct=struct, mode="exact")
    assert max_relF_batch(theta, g["theta_L"]) < 3e-5
    grads = ex.glad_backward(g["S"], p, L, tr, diag, loss_S=loss_S, struct=struct, mode="exact")
    for key in ex.PARAM_KEYS:
        ref = g["grad." + key]
        assert relF(grads[key], ref) < 1e-4 or np.abs(grads[key] - ref).max() < 2e-6, (key, grads[key], ref)


def test_exact_and_ns10_diverge_in_trained_regime():
    g = load("cell_d128_b2_L30_trained")
    p = ex.params64(g, "param.")
    theta, _ = ex.glad_forward(g["S"], p, int(g["L"]), 0, mode="exact")
    assert max_relF_batch(theta, g["theta_L"]) > 1e-4


def test_consensus():
    g = load("consensus")
    np.testing.assert_array_equal(ns.consensus_min(torch.from_numpy(g["theta_K"])).numpy(), g["out_min"])
    np.testing.assert_array_equal(ex.consensus_min(g["theta_K"]).astype(np.float32), g["out_min"])


# ----------------------------------------------------------------------------------------------- outside the comfortable regime (round 3)
REGIME = sorted(os.path.basename(p)[:-4] for p in __import__("glob").glob(os.path.join(GOLDEN, "regime_*.npz")))
REGIME_TABLE = {r["case"]: r for r in __import__("json").load(open(os.path.join(GOLDEN, "regime_sweep.json")))}
NOISE = __import__("json").load(open(os.path.join(GOLDEN, "grad_noise_floor.json")))


@pytest.mark.parametrize("name", ["regime_rawcov_d32_eo0.03_trained", "regime_nltd_d48_n30_shift0.001_trained",
                                  "regime_scaled_d32_c64_trained", "cell_d25_b1_L15_trained"])
def test_spectral_form_is_the_matrix_iteration_in_exact_arithmetic(name):
    """The reference's function -- 10 Newton-Schulz steps in MATRIX arithmetic, its hand-written 10-step backward -- evaluated in
    float64 (oracle/glad_ns.py, the NS-faithful restatement, run in double) equals the spectral form (oracle/glad_exact.py, mode
    "ns10") to 1e-10, also where cond(b^T b + 4/lam I) is 4e3 ... 2e5 and the fp32 reference is 1e-4 / 20 % away from both.  So the
    distance reference <-> spectral oracle in tests/golden/regime_sweep.json is the reference's own fp32 rounding noise, not a
    modelling difference."""
    g = load(name)
    L = int(g["L"])
    p = {k: torch.tensor(np.array(g["param." + k]), dtype=torch.float64, requires_grad=True) for k in ns.PARAM_KEYS}
    th, loss = ns.forward_uGLAD(torch.tensor(g["S"], dtype=torch.float64), p, L=L, INIT_DIAG=int(g["INIT_DIAG"]))
    loss.backward()
    p64 = ex.params64(g, "param.")
    theta, tr = ex.glad_forward(g["S"], p64, L, int(g["INIT_DIAG"]), mode="ns10")
    grads = ex.glad_backward(g["S"], p64, L, tr, int(g["INIT_DIAG"]), mode="ns10")
    assert max_relF_batch(th.detach().numpy(), theta) < 1e-12
    for key in ex.PARAM_KEYS:
        assert relF(p[key].grad.numpy(), grads[key]) < 1e-7, key  # (measured <= 2e-9)


@pytest.mark.parametrize("name", REGIME)
def test_regime_goldens_vs_spectral_oracle(name):
    """Reference-made goldens outside uGLAD's min-max-normalised input regime (N < D with small repair shifts, covariances of raw
    samples, scaled covariances, lambda driven small): Theta of the fp64 spectral oracle within 1e-4 of the reference wherever
    cond(b^T b + 4/lam I) <= 1500 (the validated bound, uglad_validated_cond()), within twice the reference's own noise beyond; the
    table's cond_max is reproduced; gradients within max(1e-4, 2 x the reference's fp32 noise)."""
    g = load(name)
    row = REGIME_TABLE[name]
    p = ex.params64(g, "param.")
    L = int(g["L"])
    theta, tr = ex.glad_forward(g["S"], p, L, int(g["INIT_DIAG"]), mode="ns10")
    err = max_relF_batch(theta, g["theta_L"])
    assert abs(err - row["theta_relF_reference_vs_fp64_spectral"]) < 1e-3 * err + 1e-10  # the committed table is this computation
    if row["cond_max"] <= 1500.0:
        assert err < 2e-5, (err, row["cond_max"])  # (measured <= 1.1e-5; the tolerance of the north star is 1e-4)
    else:
        assert err < 2e-4, (err, row["cond_max"])  # 1.04e-4 at cond 4.4e3: the reference's fp32 matrix iteration itself
    grads = ex.glad_backward(g["S"], p, L, tr, int(g["INIT_DIAG"]), mode="ns10")
    noise = max(NOISE[name]["grads"].values())
    for key in ex.PARAM_KEYS:
        assert relF(g["grad." + key], grads[key]) <= max(1e-4, 1.001 * noise), key  # (normalised by the oracle's value, as the floor was)


def test_oracle_singular_theta_gives_nan_not_an_exception():
    """torch.logdet semantics in the oracle's loss (main.py:307): det = 0 -> -inf, det < 0 -> NaN; its backward NaN, never LinAlgError."""
    th = np.zeros((2, 3, 3))
    th[1] = -np.eye(3)
    S = np.stack([np.eye(3)] * 2)
    assert not np.isfinite(ex.loss_fwd(th, S))
    G = ex.loss_bwd(th, S)
    assert np.isnan(G[0]).all() and np.isfinite(G[1]).all()
