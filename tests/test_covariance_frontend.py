"""Covariance front-end: table -> min-max normalisation -> empirical covariance -> eigenvalue repair.
Goldens: outputs of the real reference (tests/golden/make_cov_goldens.py).  CPU: the oracle restatement against the
goldens, and the unmodified kernel sources on the SIMT emulator against the goldens.  GPU: the HIP kernels through the C ABI."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import covariance as ocov

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "cov_*.npz")))
SMALL = [c for c in CASES if "d256" not in c and "d128" not in c]  # what the emulator finishes in seconds

# fp32 kernels against the reference's fp64 arithmetic: the covariance entries are O(1e-2) after min-max scaling and carry
# ~N rounding errors of 6e-8 each; the repair adds (offset - min eig) with min eig known to ~1e-7 absolute.
TOL = 2e-5  # relative Frobenius


def relerr(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    Xn = ocov.normalize_min_max(g["X"])
    if "Xn" in g.files:
        assert np.abs(Xn - g["Xn"]).max() < 1e-14
        assert np.abs(ocov.empirical_cov(Xn) - g["S_raw"]).max() < 1e-14
    S = ocov.get_covariance(Xn, offset=float(g["offset"]))
    assert relerr(S, g["S"]) < 1e-9  # (the shift uses min eig: eigenvalue routines agree to ~1e-12 absolute)


def _check_device(lib, g, device):
    X = torch.from_numpy(g["X"]).to(device)
    S = lib.covariance(X, normalize=True, eval_offset=float(g["offset"]), repair=True)
    if device != "cpu":
        torch.cuda.synchronize()
    S = S.cpu().numpy()
    assert np.isfinite(S).all()
    assert np.abs(S - np.swapaxes(S, 1, 2)).max() == 0.0  # exactly symmetric
    for k in range(S.shape[0]):
        assert relerr(S[k], g["S"][k]) < TOL, (k, relerr(S[k], g["S"][k]))
    if "S_raw" in g.files:  # and without the repair
        S0 = lib.covariance(X, normalize=True, repair=False)
        S0 = S0.cpu().numpy()
        for k in range(S.shape[0]):
            assert relerr(S0[k], g["S_raw"][k]) < TOL
    # normalize = 0 on an already normalised table gives the same matrix
    if "Xn" in g.files:
        S1 = lib.covariance(torch.from_numpy(g["Xn"].astype(np.float32)).to(device), normalize=False, repair=False).cpu().numpy()
        for k in range(S.shape[0]):
            assert relerr(S1[k], g["S_raw"][k]) < TOL


@pytest.mark.parametrize("name", SMALL)
def test_emulated_kernels_match_reference(emul, name):
    _check_device(emul, np.load(os.path.join(GOLDEN, name + ".npz")), "cpu")


def test_constant_column_gives_nan_like_reference(emul):
    X = np.random.default_rng(0).standard_normal((1, 30, 8)).astype(np.float32)
    X[0, :, 3] = 2.5
    S = emul.covariance(torch.from_numpy(X), normalize=True, repair=False).numpy()
    ref = ocov.empirical_cov(ocov.normalize_min_max(X))
    assert np.isnan(S[0, 3, :]).all() and np.isnan(ref[0, 3, :]).all()
    ok = np.isfinite(ref[0])
    assert np.abs(S[0][ok] - ref[0][ok]).max() < 1e-6


def test_argument_errors(emul):
    from uglad_amd._lib import UgladError

    with pytest.raises(UgladError):
        emul.covariance(torch.zeros(1, 4, 300), repair=False)  # D > uglad_max_dim()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_matches_reference(name):
    from uglad_amd import _lib

    _check_device(_lib.get_lib(), np.load(os.path.join(GOLDEN, name + ".npz")), "cuda")


@pytest.mark.gpu
def test_gpu_fit_with_device_covariance_matches_host_path():
    import uglad_amd

    g = np.load(os.path.join(GOLDEN, "fit_direct_d25.npz"))
    X = g["X"]
    out = []
    for dev_cov in (False, True):
        torch.manual_seed(0)
        m = uglad_amd.uGLAD_GL(device_covariance=dev_cov)
        m.fit(X, centered=False, epochs=20, lr=0.002, INIT_DIAG=0, L=15, verbose=False)
        out.append(m.precision_.copy())
    assert relerr(out[1], out[0].astype(np.float64)) < 1e-4


def test_ragged_tables_under_device_covariance(emul):
    """Missing mode hands K row-subsampled folds to the covariance front-end; with N % K != 0 they differ in length (and so
    may multitask tables).  The device path groups the tables by shape instead of stacking them into one array."""
    import uglad_amd
    from uglad_amd import main

    rng = np.random.default_rng(5)
    X = rng.standard_normal((50, 6))  # 50 % 3 != 0: train folds of 33, 33, 34 rows
    X[rng.random(X.shape) < 0.1] = np.nan
    out = []
    for dev_cov in (False, True):
        torch.manual_seed(1)
        est = uglad_amd.uGLAD_GL(device_covariance=dev_cov)
        est.fit(X.copy(), epochs=2, lr=0.01, L=3, verbose=False, k_fold=3, mode="missing")
        out.append(est.precision_.copy())
    assert relerr(out[1], out[0].astype(np.float64)) < 1e-4
    tabs = [rng.random((n, 5)) for n in (40, 31, 40)]
    with main.device_covariance(True):
        S_dev = main._covariance(tabs, 0.1).numpy()
    S_host = main._covariance(tabs, 0.1).numpy()
    assert S_dev.shape == (3, 5, 5) and relerr(S_dev, S_host.astype(np.float64)) < TOL


def test_repair_decision_near_the_threshold_follows_fp64(emul):
    """The reference repairs where the fp64 minimum eigenvalue is <= 1e-6 (prepare_data.py:347-352).  An fp32 solver cannot
    resolve that threshold; tables whose covariance has its smallest eigenvalue within the guard band of it are re-decided on
    the host in fp64, so the device path and the host path take the same branch (the shift is an O(offset) step)."""
    from uglad_amd import main
    from uglad_amd.utils import prepare_data as pd_

    rng = np.random.default_rng(9)
    N, D = 64, 6
    B = rng.random((N, D))
    tabs = []
    for eps in (3e-7, 9e-7, 1.1e-6, 3e-6, 2e-5):  # smallest eigenvalue of the covariance ~ eps: both sides of 1e-6
        Xc = B - B.mean(0)
        U, s, Vt = np.linalg.svd(Xc, full_matrices=False)
        s[-1] = np.sqrt(eps * N)
        tabs.append(U @ np.diag(s) @ Vt + B.mean(0))
    S_host = pd_.get_covariance(tabs, offset=0.1)
    with main.device_covariance(True):
        S_dev = main._covariance(tabs, 0.1).numpy()
    shifted_host = [np.linalg.eigvalsh(S).min() > 0.05 for S in S_host]
    shifted_dev = [np.linalg.eigvalsh(S.astype(np.float64)).min() > 0.05 for S in S_dev]
    assert shifted_host == shifted_dev == [True, True, False, False, False]
    for a, b in zip(S_dev, S_host):
        assert relerr(a, b) < 5e-5
