"""SURVEY.md 8f N3 / N4 -- what follows the hot path: conditional Gaussian / MAP estimate, partial correlations and the
support-recovery metrics, on the device through the C ABI, against (a) vectors captured from the REAL reference
(tests/golden/map_*.npz, metrics_k3_d20.npz; make_goldens_r2.py) and (b) the fp64 oracle on seeded inputs incl. edge cases.
The oracle itself is pinned to the goldens in the CPU part; the kernels run on the SIMT emulator (CPU) and on the GPU."""
import os

import numpy as np
import pytest
import torch

from oracle import after_path as oap

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
MAPS = ["map_d25", "map_d64"]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


# ----------------------------------------------------------------------------------------------- oracle vs reference
@pytest.mark.parametrize("name", MAPS)
def test_oracle_conditional_gaussian_matches_reference(name):
    g = load(name)
    full, cov, logp = oap.conditional_gaussian(g["precision"], g["mean"], g["observed_idx"], g["observed_values"])
    np.testing.assert_allclose(full, g["full_mean"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(cov, g["cond_cov"], rtol=1e-6, atol=1e-9)  # (the reference inverts a float32 precision_ in float32)
    assert abs(logp - np.log(float(g["pdf"]))) < 1e-6  # (its covariance went through float32 for map_d25)
    np.testing.assert_allclose(oap.map_estimate(g["precision"], g["mean"], g["observed_idx"], g["observed_values"]),
                               g["map_clipped"], rtol=1e-10, atol=1e-12)


def test_oracle_metrics_and_partial_correlations_match_reference():
    g = load("metrics_k3_d20")
    for k in range(g["true_theta"].shape[0]):
        m = oap.support_metrics(g["true_theta"][k], g["pred_theta"][k])
        np.testing.assert_array_equal(np.round(m, 3), g["metrics"][k])
        np.testing.assert_allclose(oap.partial_correlations(g["pred_theta"][k]), g["partial_correlations"][k], rtol=0, atol=1e-15)


# ----------------------------------------------------------------------------------------------- kernels
def _check_map(name):
    import uglad_amd

    g = load(name)
    full, cov, pdf = uglad_amd.conditional_gaussian_with_probabilities(g["precision"], g["mean"], g["observed_idx"],
                                                                     g["observed_values"])
    # fp32 eigensolver + one step of iterative refinement against the reference's fp64 scipy solve
    scale = np.abs(g["full_mean"]).max()
    err_mean = np.abs(full - g["full_mean"]).max() / scale
    err_cov = np.linalg.norm(cov - g["cond_cov"]) / np.linalg.norm(g["cond_cov"])
    err_logp = abs(np.log(pdf) - np.log(float(g["pdf"])))
    print(f"{name}: conditional mean {err_mean:.2e}, conditional covariance {err_cov:.2e}, log pdf {err_logp:.2e}")
    assert err_mean < 2e-6 and err_cov < 2e-6 and err_logp < 2e-4
    assert cov.shape == g["cond_cov"].shape and np.array_equal(cov, cov.T)
    obs = g["observed_idx"]
    assert np.array_equal(full[obs].astype(np.float32), g["observed_values"].astype(np.float32))  # observed values pass through

    class Fitted:
        precision_, location_ = g["precision"], g["mean"]
        node_names_ = [f"n{i}" for i in range(len(g["mean"]))]

    got = uglad_amd.compute_map_estimate({f"n{i}": float(v) for i, v in zip(obs, g["observed_values"])}, Fitted)
    assert np.abs(got - g["map_clipped"]).max() < 2e-6 and got.min() >= 0.0 and got.max() <= 1.0


def _check_map_batch_and_edges(device):
    import uglad_amd

    rng = np.random.default_rng(12)
    K, D = 5, 33
    A = rng.standard_normal((K, D, D))
    P = A @ A.transpose(0, 2, 1) / D + 0.5 * np.eye(D)
    mu = rng.random((K, D))
    mask = rng.random((K, D)) < 0.4
    mask[0] = False            # nothing observed: the mean comes back unchanged, cond_cov = P^-1
    mask[1] = True             # everything observed: the values come back, cond_cov = identity, log pdf = 0
    vals = rng.random((K, D))
    full, cov, logp = uglad_amd.conditional_gaussian_batch(P, mu, mask.astype(np.float32), vals)
    full, cov, logp = full.cpu().numpy(), cov.cpu().numpy(), logp.cpu().numpy()
    for k in range(K):
        obs = np.nonzero(mask[k])[0]
        un = np.nonzero(~mask[k])[0]
        rf, rc, rl = oap.conditional_gaussian(P[k], mu[k], obs, vals[k][obs])
        assert np.abs(full[k] - rf).max() < 5e-6, k
        assert np.linalg.norm(cov[k][np.ix_(un, un)] - rc) <= 5e-6 * max(1.0, np.linalg.norm(rc)), k
        assert np.allclose(cov[k][np.ix_(obs, obs)], np.eye(len(obs)), rtol=0, atol=1e-6), k
        assert abs(logp[k] - rl) < 2e-4 * max(1.0, abs(rl)), k
    Pbad = P[:1].copy()
    Pbad[0] = -Pbad[0]         # L_uu not positive definite: the density is NaN, nothing aborts
    _, _, lp = uglad_amd.conditional_gaussian_batch(Pbad, mu[:1], mask[2:3].astype(np.float32), vals[:1])
    assert np.isnan(lp.cpu().numpy()[0])


def _check_metrics():
    import uglad_amd

    g = load("metrics_k3_d20")
    got = uglad_amd.device_report_metrics(g["true_theta"], g["pred_theta"])
    for k, d in enumerate(got):
        assert tuple(d) == oap.METRIC_KEYS
        np.testing.assert_array_equal(np.array([d[key] for key in oap.METRIC_KEYS]), g["metrics"][k])  # the reference's 3 decimals
    pc = uglad_amd.get_partial_correlations(g["pred_theta"])
    np.testing.assert_allclose(pc, g["partial_correlations"], rtol=0, atol=2e-7)
    assert np.array_equal(uglad_amd.get_partial_correlations(g["pred_theta"][0]), pc[0])
    # a tensor on the device goes through uglad_partial_correlations (fp32), K matrices per launch
    import torch
    from uglad_amd import _lib
    lib = _lib.get_lib()
    P32 = torch.as_tensor(g["pred_theta"], dtype=torch.float32)
    if lib.require_gpu:
        pd_ = uglad_amd.get_partial_correlations(P32.cuda())
        assert pd_.is_cuda
    else:  # the emulated build of the same kernel
        pd_ = lib.partial_correlations(P32.contiguous())
    np.testing.assert_allclose(pd_.cpu().numpy(), g["partial_correlations"], rtol=0, atol=2e-7)


def _check_metrics_edges(D):
    """Seeded cases against the oracle, unrounded: ties in the scores, an empty prediction, a full prediction, no true edge."""
    from uglad_amd import _lib

    rng = np.random.default_rng(D)
    K = 5
    T = np.zeros((K, D, D), dtype=np.float32)
    G = np.zeros((K, D, D), dtype=np.float32)
    for k in range(K):
        t = np.triu((rng.random((D, D)) < 0.2), 1)
        T[k] = (t + t.T) * rng.standard_normal((D, D)) + np.eye(D)
        s = np.round(rng.random((D, D)), 1)            # many tied scores, many exact zeros
        s = np.triu(s * (rng.random((D, D)) < 0.5), 1)
        G[k] = s + s.T + np.eye(D)
    G[1] = np.eye(D)                                    # nothing predicted: FDR, precision = 0/0
    G[2] = 1.0                                          # everything predicted with ONE score: AUC = 1/2
    T[3] = np.eye(D)                                    # no true edge: the ranking metrics are undefined (NaN)
    dev = _lib.device()
    out = _lib.get_lib().support_metrics(torch.from_numpy(T).to(dev), torch.from_numpy(G).to(dev)).cpu().numpy()
    for k in range(K):
        ref = oap.support_metrics(T[k], G[k])
        np.testing.assert_allclose(out[k], ref, rtol=1e-12, atol=0, equal_nan=True, err_msg=str(k))  # (summation order of AP)
    assert out[2][10] == 0.5 and np.isnan(out[3][10]) and np.isnan(out[1][0])


@pytest.mark.parametrize("name", MAPS)
def test_emulated_map_matches_reference(emul, name):
    _check_map(name)


def test_emulated_map_batch_and_edge_cases(emul):
    _check_map_batch_and_edges("cpu")


def test_emulated_metrics_match_reference(emul):
    _check_metrics()
    _check_metrics_edges(12)
    _check_metrics_edges(37)


def test_fit_reports_device_metrics(emul):
    """fit(true_theta=...) returns the reference's metrics dict, counted on the device."""
    import uglad_amd
    from uglad_amd.utils.metrics import report_metrics_all
    from uglad_amd.utils.prepare_data import get_data

    X, P = get_data(8, (0.2, 0.4), 80, 1, eig_offset=1.0, rng=3)
    est = uglad_amd.uGLAD_GL()
    res = est.fit(X[0], true_theta=P[0], epochs=2, lr=0.01, L=3, verbose=False)
    ref = report_metrics_all(P[0], est.precision_)  # host-side numpy restatement of the same definitions
    assert list(res) == list(ref)
    np.testing.assert_array_equal(np.array(list(res.values())), np.array(list(ref.values())))  # (NaN == NaN here)


def _check_beyond_the_device_solvers(D):
    """D above uglad_max_eig_dim(): the metrics report and the conditional Gaussian take the reference's own host formulation instead of
    raising (round 3: UgladError after all epochs of fit(X, true_theta=...)) -- checked against the oracle's restatement."""
    import uglad_amd
    from uglad_amd import _lib, main as um

    assert D > _lib.get_lib().max_eig_dim
    rng = np.random.default_rng(D)
    t = np.triu(rng.random((D, D)) < 0.05, 1)
    T = ((t + t.T) * rng.standard_normal((D, D)) + np.eye(D)).astype(np.float32)
    s = np.triu(np.round(rng.random((D, D)), 2) * (rng.random((D, D)) < 0.1), 1)
    G = (s + s.T + np.eye(D)).astype(np.float32)
    got = um.device_report_metrics(T, G)
    ref = oap.support_metrics(T, G)
    assert tuple(got[0]) == oap.METRIC_KEYS
    np.testing.assert_allclose(np.array([got[0][k] for k in oap.METRIC_KEYS]), np.round(ref, 3), rtol=0, atol=1e-12, equal_nan=True)
    # conditional Gaussian: a diagonally dominant precision matrix, a third of the coordinates observed
    A = rng.standard_normal((D, D)) * 0.05
    P = A @ A.T + np.eye(D)
    mean = rng.random(D)
    obs = np.sort(rng.choice(D, D // 3, replace=False))
    vals = rng.random(obs.size)
    full, cov, pdf = uglad_amd.conditional_gaussian_with_probabilities(P, mean, obs, vals)
    rfull, rcov, rlog = oap.conditional_gaussian(P, mean, obs, vals)  # (the oracle returns the LOG density)
    np.testing.assert_allclose(full, rfull, rtol=0, atol=2e-6)
    np.testing.assert_allclose(cov, rcov, rtol=0, atol=2e-6)
    assert abs(np.log(pdf) - rlog) < 1e-4 * max(1.0, abs(rlog)), (pdf, rlog)


def test_emulated_after_path_beyond_the_device_solvers(emul):
    from uglad_amd import _lib

    _check_beyond_the_device_solvers(_lib.get_lib().max_eig_dim + 9)


@pytest.mark.gpu
def test_gpu_after_path_beyond_the_device_solvers():
    _check_beyond_the_device_solvers(300)


@pytest.mark.gpu
def test_gpu_fit_beyond_256_reports_metrics():
    """fit(X, true_theta=...) on a table of 288 columns: the epochs run on the matrix-iteration path and the report comes back."""
    import uglad_amd
    from uglad_amd.utils.metrics import report_metrics_all
    from uglad_amd.utils.prepare_data import get_data

    X, P = get_data(288, (0.02, 0.04), 600, 1, eig_offset=1.0, rng=11)
    est = uglad_amd.uGLAD_GL()
    res = est.fit(X[0], true_theta=P[0], epochs=2, lr=0.01, L=3, verbose=False)
    ref = report_metrics_all(P[0], est.precision_)
    assert list(res) == list(ref)
    np.testing.assert_array_equal(np.array(list(res.values())), np.array(list(ref.values())))
    assert est.precision_.shape == (288, 288) and np.isfinite(est.precision_).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", MAPS)
def test_gpu_map_matches_reference(name):
    _check_map(name)


@pytest.mark.gpu
def test_gpu_map_batch_metrics_and_edge_cases():
    _check_map_batch_and_edges("cuda")
    _check_metrics()
    for D in (12, 37, 129, 256):
        _check_metrics_edges(D)
