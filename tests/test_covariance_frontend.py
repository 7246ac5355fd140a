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
