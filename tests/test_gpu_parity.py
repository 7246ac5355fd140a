"""GPU parity suite (pytest -m gpu): the HIP path through the C ABI against
  (a) the reference-captured goldens (tests/golden), every shape incl. the D=64 / D=128 subsamples of configs 2-3,
  (b) the oracle on seeded inputs at sizes it finishes in seconds,
  (c) size-independent properties at BASELINE.json's full sizes (determinism, symmetry, M-invariance of lambda-free
      quantities, consistency of forward-only vs training forward).
Tolerance from the north star: precision matrices within 1e-4 relative Frobenius of the reference."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from oracle import glad_exact as ex
from oracle import glad_ns as ns

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CELLS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "cell_*.npz")))
TOL = 1e-4
FIT_TOL = 1e-4  # end-to-end precision_ of the fit goldens: the same contract (observed 6e-7 ... 2e-5)


def relF(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def max_relF(a, b):
    return max(relF(a[i], b[i]) for i in range(b.shape[0]))


@pytest.fixture(scope="module")
def lib():
    from uglad_amd import _lib

    h = _lib.get_lib()
    assert h.path.endswith("libuglad_hip.so") and h.require_gpu  # the native gfx950 build, nothing else
    return h


def load_model(g, prefix="param."):
    import uglad_amd

    m = uglad_amd.GladParams(1.0, device="cuda")
    m.load_state_dict({k: torch.from_numpy(np.array(g[prefix + k])) for k in ex.PARAM_KEYS})
    return m


_NOISE_FILE = os.path.join(GOLDEN, "grad_noise_floor.json")
_NOISE = json.load(open(_NOISE_FILE))
GRAD_CONTRACT = 1e-4


def grad_tolerance(name, key=None, whole_golden=False):
    """max(contract, 2 x the reference's OWN fp32 noise on this golden AND THIS TENSOR): a fixed bound, never refreshed from the build
    under test.  The noise (tests/golden/grad_noise_floor.json, made by tests/golden/measure_grad_noise.py in the build container) is the
    distance of the reference's fp32 gradient from the fp64 evaluation of the same function.  Round 3 took the golden's worst tensor for all
    eleven, which let a regression of 100x in an accurate tensor pass on the noisy goldens; since round 4 every tensor has its own bound
    (profiles/r04_grad_vs_fp64.txt: all 36 goldens x 11 tensors pass but one).  `whole_golden` keeps the old rule for the goldens OUTSIDE
    uglad_validated_cond() (cond 4.4e3 ... 2.3e5, where fit() / predict() warn): there the function itself is ill-conditioned -- one forward
    perturbation moves all 42 gradients -- and the kernel's theta_init_offset on regime_nltd_d48_n30_shift0.001 (cond 2.3e5) is 7.6e-3 from the
    fp64 value where the reference's is 1.2e-3 and its worst tensor 0.2."""
    rec = _NOISE.get(name)
    if rec is None:
        return GRAD_CONTRACT
    noise = max(rec["grads"].values()) if (whole_golden or key is None) else rec["grads"][key]
    return max(GRAD_CONTRACT, 2.0 * noise)


def fp64_grad_tolerance(name):
    """Fixed bound on the kernels' distance from the fp64 evaluation of the reference's function (oracle/glad_exact.py, mode ns10), per golden
    class -- the kernels' OWN accuracy, which the comparison with a noisy reference cannot see.  Written down from profiles/r04_grad_vs_fp64.txt
    (scripts/grad_vs_fp64.py on MI355X, worst of the 11 tensors): D <= 129 at most 2.2e-5 -> the 1e-4 contract itself, against the exact value;
    fresh parameters at D = 256 / 288 / 1024 at most 4.6e-7 -> 1e-4; trained parameters at D = 200 / 256 / 320: 1.4e-4 / 8.6e-6 / 7.7e-5 (the
    reference: 1.2e-3 / 1.1e-4 / 9.9e-4) -> 3e-4; cell_d512 (Theta_L indefinite with 104 negative eigenvalues, cond 2e5; the reference is 2.4e-2
    from its own exact value): 8.4e-3 -> 2e-2."""
    if name == "cell_d512_b1_L15_trained":
        return 2e-2
    return 1e-4 if (_cell_dim(name) <= 129 or "fresh" in name) else 3e-4


def tiny_gradient(got, ref, grads_ref):
    """The escape for a tensor whose reference gradient is (numerically) zero: absolute difference below 1e-6 of the LARGEST entry of the 42
    reference gradients (round 3 had an unscaled 1e-6)."""
    scale = max(float(np.abs(np.asarray(v)).max()) for v in grads_ref.values())
    return float(np.abs(np.asarray(got) - np.asarray(ref)).max()) < 1e-6 * scale


def oracle_fp64(g, prefix="param."):
    """fp64 evaluation of the reference's function on a golden's inputs: Theta_L and the 42 gradients (the checker; CPU)."""
    p = ex.params64(g, prefix)
    L, diag = int(g["L"]), int(g["INIT_DIAG"])
    kw = {"loss_S": g["loss_S"] if "loss_S" in g else None, "struct": g["struct"] if "struct" in g else None}
    theta, tr = ex.glad_forward(g["S"], p, L, diag, mode="ns10", **kw)
    return theta, ex.glad_backward(g["S"], p, L, tr, diag, mode="ns10", **kw)


def assert_gradients(name, g, sd, whole_golden=False, fp64_tol=None, grads64=None):
    """Every parameter tensor against the reference's golden (per-tensor bound) and, with fp64_tol, against the fp64 oracle."""
    ref_all = {key: g["grad." + key] for key in ex.PARAM_KEYS}
    observed = {}
    for key in ex.PARAM_KEYS:
        got = sd[key].grad.cpu().numpy()
        observed[key] = relF(got, ref_all[key])
        assert observed[key] < grad_tolerance(name, key, whole_golden) or tiny_gradient(got, ref_all[key], ref_all), \
            (name, key, "vs reference", observed[key], grad_tolerance(name, key, whole_golden))
        if fp64_tol is not None:
            e64 = relF(got, grads64[key])
            assert e64 < fp64_tol or tiny_gradient(got, grads64[key], grads64), (name, key, "vs fp64 oracle", e64, fp64_tol)
    return observed


def record_grad_errors(name, observed, theta_err):
    """Side output, for the record only (profiles/r03_grad_errors_observed.json): gpurun_out/grad_errors_observed.json (merged back by gpurun)."""
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, "grad_errors_observed.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[name] = {"theta_relF": theta_err, "grads": observed}
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def trained_model():
    return load_model(np.load(os.path.join(GOLDEN, "params_trained.npz")), "")


class spectral_path:
    """Within the block the eigensolver-based cell runs wherever it exists (D <= 256): the library's automatic choice hands few large
    matrices (128 < D <= 256) to the matrix-iteration path, and these tests are about the spectral kernels."""

    def __init__(self, lib):
        self.lib = lib

    def __enter__(self):
        self.lib.set_matrix_iteration(0)

    def __exit__(self, *exc):
        self.lib.set_matrix_iteration(-1)


# ----------------------------------------------------------------------------------------------- solver
@pytest.mark.parametrize("D", [1, 2, 3, 7, 16, 17, 25, 31, 32, 33, 64, 100, 128, 129, 200, 256, (7, "wg"), (25, "wg"), (32, "wg")])
def test_symeig(lib, D, monkeypatch):
    import uglad_amd

    if isinstance(D, tuple):  # D <= 32 goes through the one-wave tridiagonalisation (tridiag_wave.h); the workgroup kernel stays covered
        D = D[0]
        monkeypatch.setenv("UGLAD_TRIDIAG_WAVE", "0")
    torch.manual_seed(D)
    M = 37 if D <= 128 else 9
    A = torch.randn(M, D, D, device="cuda")
    A = (A + A.transpose(1, 2)).contiguous()
    A[1] = torch.diag((torch.arange(D, device="cuda") % 3).float())  # degenerate, already diagonal
    A[2] = 0.0
    if D > 4:
        blk = torch.randn(4, 4, device="cuda")
        A[3] = torch.block_diag(blk + blk.T, torch.eye(D - 4, device="cuda"))  # clustered spectrum
    beta, U = uglad_amd.batch_symeig(A)
    if D <= 128:  # independent on-device cross-check: the Jacobi solver finds the same spectrum
        Uj, bj = torch.empty_like(A), torch.empty(M, D, device="cuda")
        lib.symeig(A, Uj, bj, jacobi=True)
        sc = beta.abs().max(dim=1).values.clamp_min(1.0)
        assert ((bj.sort(dim=1).values - beta).abs().max(dim=1).values / sc).max().item() < 5e-6
    rec = (U * beta[:, None, :]) @ U.transpose(1, 2)
    scale = A.flatten(1).norm(dim=1).clamp_min(1.0)
    assert ((rec - A).flatten(1).norm(dim=1) / scale).max().item() < 5e-6
    eye = torch.eye(D, device="cuda")
    assert (U.transpose(1, 2) @ U - eye).abs().max().item() < 5e-6
    w = torch.linalg.eigvalsh(A.double())
    assert ((beta.double().sort(dim=1).values - w).abs().max(dim=1).values / w.abs().max(dim=1).values.clamp_min(1.0)).max() < 5e-6


def test_symeig_every_size_up_to_64(lib):
    """Every D the small kernels see (the one-wave tridiagonalisation up to 32, 128 NT threads beyond), three matrices each: reconstruction,
    orthogonality, spectrum against float64."""
    import uglad_amd

    for D in range(1, 65):
        g = torch.Generator(device="cpu").manual_seed(1000 + D)
        A = torch.randn(3, D, D, generator=g)
        A = (A + A.transpose(1, 2))
        A[2] = A[2] * 1e-3 + torch.diag(torch.linspace(-2.0, 2.0, D))  # nearly diagonal: tiny reflectors
        A = A.cuda().contiguous()
        beta, U = uglad_amd.batch_symeig(A)
        rec = (U * beta[:, None, :]) @ U.transpose(1, 2)
        scale = A.flatten(1).norm(dim=1).clamp_min(1.0)
        assert ((rec - A).flatten(1).norm(dim=1) / scale).max().item() < 5e-6, D
        assert (U.transpose(1, 2) @ U - torch.eye(D, device="cuda")).abs().max().item() < 5e-6, D
        w = torch.linalg.eigvalsh(A.double())
        assert ((beta.double().sort(dim=1).values - w).abs().max(dim=1).values / w.abs().max(dim=1).values.clamp_min(1.0)).max() < 5e-6, D


@pytest.mark.parametrize("D", [3, 16, 33, 128])
def test_degenerate_and_nearly_degenerate_spectra_vs_fp64_oracle(lib, D):
    """Covariances whose b = S/lambda - Z has repeated or nearly repeated eigenvalues -- exactly diagonal, the identity, equicorrelation (D - 1 equal
    eigenvalues), repeated 4 x 4 blocks, AR(1), nearly diagonal, nearly the identity -- where an eigenvector-based matrix function and its
    divided differences can fail while every random input passes: Theta_L and the 42 gradients against the fp64 oracle of the same function.
    Observed over D = 2 ... 128 (scripts/structured_cell_probe.py, profiles/r04_structured_cell_probe.txt): Theta <= 8e-7 (equicorrelation at
    D = 100: 3.8e-5, where the reference itself is 1.9e-3 from the fp64 value), gradients <= 3.1e-5."""
    import uglad_amd

    pz = np.load(os.path.join(GOLDEN, "params_trained.npz"))
    i = np.arange(D)
    rng = np.random.default_rng(D)
    R = rng.standard_normal((D, D))
    R = R + R.T
    blk = np.eye(D)
    for b in range(D // 4):
        blk[4 * b:4 * b + 4, 4 * b:4 * b + 4] = 0.6 * np.eye(4) + 0.4 * np.ones((4, 4))
    kinds = {"diagonal": np.diag(np.linspace(0.5, 2.0, D)), "identity": np.eye(D), "equicorrelation": 0.5 * np.eye(D) + 0.5 * np.ones((D, D)),
             "blocks": blk, "AR(1)": 0.7 ** np.abs(i[:, None] - i[None, :]), "near-diagonal": np.diag(np.linspace(0.5, 2.0, D)) + 1e-3 * R,
             "near-identity": np.eye(D) + 1e-4 * R}
    model = uglad_amd.GladParams(1.0, device="cuda")
    model.load_state_dict({k: torch.from_numpy(np.array(pz[k])) for k in pz.files})
    p64 = ex.params64({k: pz[k] for k in pz.files})
    L = 8
    for name, S64 in kinds.items():
        S = np.ascontiguousarray(S64[None].astype(np.float32))
        for prm in model.parameters():
            prm.grad = None
        theta, loss = uglad_amd.forward_uGLAD(torch.from_numpy(S).cuda(), model, L=L, INIT_DIAG=0)
        loss.backward()
        th64, tr = ex.glad_forward(S.astype(np.float64), p64, L, 0, mode="ns10")
        g64 = ex.glad_backward(S.astype(np.float64), p64, L, tr, 0, mode="ns10")
        sd = dict(model.named_parameters())
        got = np.concatenate([sd[k].grad.cpu().numpy().astype(np.float64).reshape(-1) for k in ex.PARAM_KEYS])
        ref = np.concatenate([np.asarray(g64[k], np.float64).reshape(-1) for k in ex.PARAM_KEYS])
        assert np.isfinite(loss.item()), (D, name)
        assert relF(theta.detach().cpu().numpy(), th64) < (1e-4 if name == "equicorrelation" else 1e-5), (D, name, relF(theta.detach().cpu().numpy(), th64))
        assert relF(got, ref) < 3e-4, (D, name, relF(got, ref))


def test_dimension_limits(lib):
    import uglad_amd
    from uglad_amd._lib import UgladError

    with pytest.raises(UgladError):
        uglad_amd.batch_symeig(torch.zeros(1, lib.max_dim + 1, lib.max_dim + 1, device="cuda"))
    with pytest.raises(UgladError):
        uglad_amd.batch_symeig(torch.zeros(1, 8, 8))  # host tensor: no CPU fallback


# ----------------------------------------------------------------------------------------------- goldens
@pytest.mark.parametrize("name", CELLS)
def test_forward_backward_vs_reference_goldens(lib, name):
    import uglad_amd

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = load_model(g)
    S = torch.from_numpy(g["S"]).cuda()
    kw = {}
    if "loss_S" in g:
        kw["loss_Sb"] = torch.from_numpy(g["loss_S"]).cuda()
    if "struct" in g:
        kw["struct_theta"] = torch.from_numpy(g["struct"]).cuda()
    theta, loss = uglad_amd.forward_uGLAD(S, model, L=int(g["L"]), INIT_DIAG=int(g["INIT_DIAG"]), **kw)
    loss.backward()
    err = max_relF(theta.detach().cpu().numpy(), g["theta_L"])
    print(f"{name}: Theta rel-Frobenius vs reference {err:.2e}, loss {loss.item():.6f} vs {float(g['loss']):.6f}")
    assert err < TOL, err
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * max(1.0, abs(float(g["loss"])))
    sd = dict(model.named_parameters())
    # Gradient contract: <= 1e-4 relative per parameter tensor -- or, where the reference's own fp32 gradient OF THAT TENSOR is
    # noisier than that, twice its noise -- and, against the fp64 value of the same function, the fixed bound of this golden's class.
    _, grads64 = oracle_fp64(g)
    observed = assert_gradients(name, g, sd, fp64_tol=fp64_grad_tolerance(name), grads64=grads64)
    record_grad_errors(name, observed, err)
    assert torch.equal(theta, theta.transpose(1, 2))  # exactly symmetric by construction


def test_backward_pass_in_one_launch_equals_one_launch_per_step(lib, monkeypatch):
    """D <= 128: the L steps of the backward pass share ONE launch (dL/dZ stays in LDS, the rhoNN gradient sums in registers);
    UGLAD_PERSISTENT_BWD=0 launches one kernel per step.  Same dL/dZ chain, so the gradients differ only by the order in which the 28 sums
    are added up; with a single step the two are the same code path."""
    import uglad_amd
    from uglad_amd.utils.prepare_data import synthetic_covariance_batch

    for D, B, L in ((33, 5, 4), (128, 9, 6), (64, 3, 1)):
        S = torch.from_numpy(synthetic_covariance_batch(B, D, seed=50 + D)).cuda()
        W = torch.randn(B, D, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
        grads = []
        for flag in ("1", "0"):
            monkeypatch.setenv("UGLAD_PERSISTENT_BWD", flag)
            model = trained_model()
            (uglad_amd.glad(S, model, L=L) * W).sum().backward()
            grads.append(torch.cat([p.grad.reshape(-1) for p in model.parameters()]))
        scale = float(grads[1].abs().max())
        assert torch.allclose(grads[0], grads[1], rtol=0, atol=3e-6 * scale), ((grads[0] - grads[1]).abs().max(), scale)
        if L == 1:
            assert torch.equal(grads[0], grads[1])


# ----------------------------------------------------------------------------------------------- beyond the eigensolver: the matrix iteration
@pytest.mark.parametrize("D,B,L", [(1, 2, 2), (2, 3, 3), (3, 1, 3), (7, 3, 4), (31, 2, 3), (32, 2, 3), (33, 2, 3), (64, 2, 5), (100, 2, 5), (129, 2, 4),
                                   (256, 1, 6)])
def test_matrix_iteration_path_equals_spectral_path(lib, D, B, L):
    """csrc/wide_ns.h (the reference's Newton-Schulz iteration as dense tile products, what D > 256 runs on) forced at sizes the spectral
    path serves too: same Theta, loss and 42 gradients up to the fp32 noise of ten matrix iterations; the cond diagnostic is the
    Gershgorin upper bound of what the spectral path reports."""
    import uglad_amd
    from uglad_amd.utils.prepare_data import synthetic_covariance_batch

    S = torch.from_numpy(synthetic_covariance_batch(B, D, seed=40 + D)).cuda()
    out = []
    for forced in (0, 1):  # 0: the spectral path wherever it exists
        lib.set_matrix_iteration(forced)
        try:
            model = trained_model()
            with uglad_amd.regime_monitor() as mon:
                theta, loss = uglad_amd.forward_uGLAD(S, model, L=L)
            loss.backward()
        finally:
            lib.set_matrix_iteration(-1)
        out.append((theta.detach(), loss.item(), torch.cat([p.grad.reshape(-1) for p in model.parameters()]), mon.result()))
    (t0, l0, g0, c0), (t1, l1, g1, c1) = out
    assert max_relF(t1.cpu().numpy(), t0.cpu().numpy()) < 2e-5
    assert abs(l1 - l0) < 1e-5 * max(1.0, abs(l0))
    assert ((g1 - g0).abs().max() / g0.abs().max()).item() < 1e-4
    assert torch.equal(t1, t1.transpose(1, 2))
    assert c0 * 0.999 <= c1, (c0, c1)
    if D >= 4:  # (the bound takes 4 / lambda for the smallest eigenvalue: at D = 1, where the true ratio is 1, it says 74)
        assert c1 < 64 * c0, (c0, c1)


def test_matrix_iteration_batch_and_groups(lib):
    """D = 300 (beyond the eigensolver): a batch of three equals three single calls given the same lambda sequence -- here via groups,
    each matrix its own group, which is also the grouped entry on this path."""
    import uglad_amd
    from uglad_amd.glad.glad import glad_grouped
    from uglad_amd.utils.prepare_data import synthetic_covariance_batch

    D, L, G = 300, 3, 3
    S = torch.from_numpy(synthetic_covariance_batch(G, D, seed=9)).cuda()
    models = []
    for g in range(G):
        torch.manual_seed(200 + g)
        models.append(uglad_amd.GladParams(1.0 + 0.1 * g, device="cuda"))
    P = torch.stack([m.packed().detach() for m in models]).requires_grad_(True)
    W = torch.randn(G, D, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    th = glad_grouped(S, P, L=L)
    (th * W).sum().backward()
    for g in range(G):
        t1 = uglad_amd.glad(S[g:g + 1].contiguous(), models[g], L=L)
        (t1 * W[g:g + 1]).sum().backward()
        assert torch.equal(t1.detach(), th[g:g + 1].detach())
        g1 = torch.cat([p.grad.reshape(-1) for p in models[g].parameters()])
        assert torch.allclose(g1, P.grad[g], rtol=0, atol=1e-6 * float(g1.abs().max())), (g, (g1 - P.grad[g]).abs().max())


@pytest.mark.parametrize("D,B", [(257, 1), (300, 2), (510, 1), (511, 1), (384, 5), (1000, 1), (1100, 1), (2048, 1)])
def test_matrix_iteration_ragged_sizes_vs_oracle(lib, D, B):
    """Sizes that are not multiples of the tiles (odd D: scalar operand loads; 510: 16-byte loads with a ragged last tile; 384 x 5: the
    64 x 64 tiling, others the 32 x 32 one), two steps forward + backward against the fp64 oracle.  1100 and 2048 (round 4: uglad_max_dim() = 2048):
    the second layout of the L D L^T factorisation (slabs of 2048 x 2049), 2048 the largest size there is."""
    import uglad_amd
    from uglad_amd.utils.prepare_data import synthetic_covariance_batch

    g = np.load(os.path.join(GOLDEN, "params_trained.npz"))
    model = load_model(g, "")
    Snp = synthetic_covariance_batch(B, D, 4 * D if D > 512 else None, seed=D)  # (the default of 1024 samples would leave S rank-deficient at D = 1000, and S + tI near-singular with the trained offset)
    theta, loss = uglad_amd.forward_uGLAD(torch.from_numpy(Snp).cuda(), model, L=2)
    loss.backward()
    p = ex.params64(g, "")
    ref, tr = ex.glad_forward(Snp, p, 2, 0, mode="ns10")
    err = max_relF(theta.detach().cpu().numpy(), ref)
    # (D = 2048: Theta_0 is an fp32 inverse of a 2048 x 2048 matrix, 9e-6 from the fp64 inverse after its two Newton steps -- kappa x eps --
    # and Theta_L inherits it: 1.9e-5, inside the 1e-4 contract with the margin a reference at that size would need as well)
    assert err < (5e-5 if D > 1536 else 5e-6), err
    assert abs(loss.item() - tr["loss"]) < 2e-5 * abs(tr["loss"])
    grads = ex.glad_backward(Snp, p, 2, tr, 0, mode="ns10")
    sd = dict(model.named_parameters())
    print(f"D={D} B={B}: Theta vs fp64 oracle {err:.2e}; gradients " + ", ".join(f"{relF(sd[k].grad.cpu().numpy(), grads[k]):.1e}" for k in ex.PARAM_KEYS))
    for key in ex.PARAM_KEYS:
        # the shift's gradient -<G_0^T, Theta_0^2> is an fp32 inner product of 260 k cancelling terms behind an fp32 inverse: 1.7e-4 at
        # D = 510 (the reference's own fp32 value of this tensor is 1.9e-2 from fp64 at D = 512, tests/golden/grad_noise_floor.json)
        # ... and at D = 2048 every gradient passes through that fp32 inverse of a 2048 x 2048 matrix (9e-6 from the fp64 inverse): 3e-4 all round
        tol = 1e-3 if (key == "theta_init_offset" and D > 384) else (3e-4 if D > 1536 else 1e-4)
        assert relF(sd[key].grad.cpu().numpy(), grads[key]) < tol, key
    assert torch.equal(theta, theta.transpose(1, 2))


@pytest.mark.parametrize("D,B", [(130, 1), (300, 2)])
def test_matrix_iteration_diagonal_initialisation(lib, D, B):
    """INIT_DIAG = 1 (Theta_0 from the diagonal of S alone, glad.py:109-112) on the matrix-iteration path vs the fp64 oracle."""
    import uglad_amd
    from uglad_amd.utils.prepare_data import synthetic_covariance_batch

    g = np.load(os.path.join(GOLDEN, "params_trained.npz"))
    model = load_model(g, "")
    Snp = synthetic_covariance_batch(B, D, seed=D + 1)
    theta, loss = uglad_amd.forward_uGLAD(torch.from_numpy(Snp).cuda(), model, L=2, INIT_DIAG=1)
    loss.backward()
    p = ex.params64(g, "")
    ref, tr = ex.glad_forward(Snp, p, 2, 1, mode="ns10")
    grads = ex.glad_backward(Snp, p, 2, tr, 1, mode="ns10")
    assert max_relF(theta.detach().cpu().numpy(), ref) < 5e-6
    assert abs(loss.item() - tr["loss"]) < 2e-5 * abs(tr["loss"])
    sd = dict(model.named_parameters())
    for key in ex.PARAM_KEYS:
        assert relF(sd[key].grad.cpu().numpy(), grads[key]) < 1e-4, key


def test_matrix_iteration_structure_penalty_and_separate_loss_matrix(lib):
    """Beyond the eigensolver with the two loss variants of the drivers: the log-cosh structure penalty (main.py:317-333) and a loss taken
    on ONE other covariance matrix against K precision matrices (the missing-data call, main.py:620-622) -- vs the fp64 oracle."""
    import uglad_amd
    from uglad_amd.utils.prepare_data import synthetic_covariance_batch

    D, K, L = 300, 2, 2
    g = np.load(os.path.join(GOLDEN, "params_trained.npz"))
    p = ex.params64(g, "")
    Snp = synthetic_covariance_batch(K, D, seed=77)
    rng = np.random.default_rng(5)
    struct = (rng.random((K, D, D)) < 0.1).astype(np.float32)
    struct = np.maximum(struct, struct.transpose(0, 2, 1))
    for kw_k, kw_o in (({"struct_theta": torch.from_numpy(struct).cuda()}, {"struct": struct}),
                       ({"loss_Sb": torch.from_numpy(Snp[:1]).cuda()}, {"loss_S": Snp[:1]})):
        model = load_model(g, "")
        theta, loss = uglad_amd.forward_uGLAD(torch.from_numpy(Snp).cuda(), model, L=L, **kw_k)
        loss.backward()
        ref, tr = ex.glad_forward(Snp, p, L, 0, mode="ns10", **kw_o)
        grads = ex.glad_backward(Snp, p, L, tr, 0, mode="ns10", **kw_o)
        assert max_relF(theta.detach().cpu().numpy(), ref) < 5e-6
        assert abs(loss.item() - tr["loss"]) < 2e-5 * abs(tr["loss"]), (loss.item(), tr["loss"])
        sd = dict(model.named_parameters())
        for key in ex.PARAM_KEYS:
            assert relF(sd[key].grad.cpu().numpy(), grads[key]) < 1e-4, (key, list(kw_o))


def test_matrix_iteration_nan_input(lib):
    """A NaN in S: a NaN loss, no hang, no exception."""
    import uglad_amd

    D = 260
    S = (torch.eye(D, device="cuda") * 2.0)[None].contiguous()
    S[0, 3, 5] = S[0, 5, 3] = float("nan")
    theta, loss = uglad_amd.forward_uGLAD(S, trained_model(), L=2)
    assert torch.isnan(loss)


def test_matrix_iteration_limits_and_nan(lib):
    import uglad_amd
    from uglad_amd._lib import UgladError

    D = lib.max_eig_dim + 1
    S = (torch.eye(D, device="cuda") * 2.0)[None].contiguous()
    with pytest.raises(UgladError):  # the iteration IS the ten-step square root: no "exact" mode there
        uglad_amd.glad(S, trained_model(), L=1, sqrt_mode="exact")
    with pytest.raises(UgladError):
        uglad_amd.glad(torch.eye(lib.max_dim + 1, device="cuda")[None].contiguous(), trained_model(), L=1)
    lib.set_matrix_iteration(1)  # more matrices than the path's launches can number: an error, not a failed launch
    try:
        many = (torch.eye(2, device="cuda") * 2.0)[None].repeat(65535 // 3 + 1, 1, 1).contiguous()
        with pytest.raises(UgladError):
            uglad_amd.glad(many, trained_model(), L=1)
        theta = uglad_amd.glad(many[:65535 // 3].contiguous(), trained_model(), L=1)  # (the largest batch itself runs)
        assert torch.isfinite(theta).all() and torch.equal(theta[0], theta[-1])
    finally:
        lib.set_matrix_iteration(-1)
    with pytest.raises(UgladError):
        uglad_amd.batch_symeig(S)
    # diagonal input: Theta stays diagonal and the loss is the closed form
    theta, loss = uglad_amd.forward_uGLAD(S, trained_model(), L=2)
    off = theta[0] - torch.diag(torch.diagonal(theta[0]))
    assert off.abs().max().item() == 0.0
    d = torch.diagonal(theta[0]).double()
    assert abs(loss.item() - float(-torch.log(d).sum() + 2.0 * d.sum())) < 1e-4 * abs(loss.item())
    # not positive definite: NaN, not an exception, not a finite number
    th = -torch.eye(D, device="cuda")[None].contiguous()
    assert torch.isnan(uglad_amd.loss_uGLAD(th, torch.eye(D, device="cuda")[None]))


@pytest.mark.parametrize("name", [c for c in CELLS if 128 < int(__import__("re").search(r"_d(\d+)_", c).group(1)) <= 256])
def test_goldens_between_128_and_256_on_the_spectral_path(lib, name):
    """The goldens with few matrices of 128 < D <= 256 run on the matrix-iteration path when the library chooses (the test above); here
    the eigensolver-based cell is held to the same reference outputs."""
    import uglad_amd

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = load_model(g)
    S = torch.from_numpy(g["S"]).cuda()
    with spectral_path(lib):
        theta, loss = uglad_amd.forward_uGLAD(S, model, L=int(g["L"]), INIT_DIAG=int(g["INIT_DIAG"]))
        loss.backward()
    assert max_relF(theta.detach().cpu().numpy(), g["theta_L"]) < TOL
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * max(1.0, abs(float(g["loss"])))
    sd = dict(model.named_parameters())
    _, grads64 = oracle_fp64(g)
    assert_gradients(name, g, sd, fp64_tol=fp64_grad_tolerance(name), grads64=grads64)


# ----------------------------------------------------------------------------------------------- outside the comfortable regime
REGIME = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "regime_*.npz")))
REGIME_TABLE = {r["case"]: r for r in json.load(open(os.path.join(GOLDEN, "regime_sweep.json")))}


@pytest.mark.parametrize("name", REGIME)
def test_regime_goldens_parity_inside_the_bound_and_warning_outside(lib, name):
    """Reference-made goldens outside uGLAD's min-max-normalised input regime (tests/golden/make_goldens_r3.py): N < D with repair shifts
    down to 0.001, covariances of raw samples (cond(S) up to 255), scaled covariances, lambda driven small.  cond(b^T b + 4/lam I) over
    the pass -- the kernels' cond_max diagnostic -- runs from 1.3 to 2.3e5.
      * the diagnostic equals the oracle's number;
      * Theta within 2e-5 of the fp64 evaluation of the reference's function everywhere, and within the north-star 1e-4 of the REFERENCE
        wherever cond <= uglad_validated_cond() (beyond: within twice the reference's own fp32 noise -- at cond 4.4e3 its matrix iteration is
        1.04e-4 from its own exact value);
      * gradients within max(1e-4, 2 x the reference's fp32 noise on this golden);
      * predict(S=...) warns (UgladRegimeWarning) exactly when the bound is exceeded."""
    import warnings

    import uglad_amd
    from uglad_amd.glad import glad as gmod

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    row = REGIME_TABLE[name]
    model = load_model(g)
    S = torch.from_numpy(g["S"]).cuda()
    L = int(g["L"])
    with gmod.regime_monitor() as mon:
        theta, loss = uglad_amd.forward_uGLAD(S, model, L=L, INIT_DIAG=int(g["INIT_DIAG"]))
    loss.backward()
    cond = mon.result()
    assert abs(cond - row["cond_max"]) < 2e-2 * row["cond_max"], (cond, row["cond_max"])
    p64 = ex.params64(g, "param.")
    ref64, _ = ex.glad_forward(g["S"], p64, L, int(g["INIT_DIAG"]), mode="ns10")
    th = theta.detach().cpu().numpy()
    err_o, err_r = max_relF(th, ref64), max_relF(th, g["theta_L"])
    sd = dict(model.named_parameters())
    observed = {key: relF(sd[key].grad.cpu().numpy(), g["grad." + key]) for key in ex.PARAM_KEYS}
    worst = max(observed, key=observed.get)
    print(f"{name}: cond {cond:.3g}; Theta vs fp64 oracle {err_o:.2e}, vs reference {err_r:.2e} (the reference's own noise "
          f"{row['theta_relF_reference_vs_fp64_spectral']:.2e}); worst gradient {worst} {observed[worst]:.2e} (tolerance {grad_tolerance(name, worst, cond > lib.validated_cond):.2e})")
    record_grad_errors(name, observed, err_r)
    assert err_o < 2e-5
    bound = lib.validated_cond
    if cond <= bound:
        assert err_r < TOL
    else:
        assert err_r < max(TOL, 2.0 * row["theta_relF_reference_vs_fp64_spectral"])
    # gradients: per tensor inside the validated bound, the golden-wide rule beyond it (grad_tolerance's docstring); the same bounds hold
    # against the fp64 oracle -- on these goldens kernel and reference share the systematic part of the fp32 effect (lambdaNN's first weight on
    # the scaled covariances: both 9.2e-2 from the fp64 value and 3.4e-5 from each other), so the fp64 bound cannot be tighter than the noise
    _, grads64 = oracle_fp64(g)
    for key in ex.PARAM_KEYS:
        got = sd[key].grad.cpu().numpy()
        tol = grad_tolerance(name, key, whole_golden=cond > lib.validated_cond)
        assert observed[key] < tol or tiny_gradient(got, g["grad." + key], {k: g["grad." + k] for k in ex.PARAM_KEYS}), (key, observed[key], tol)
        assert relF(got, grads64[key]) < tol or tiny_gradient(got, grads64[key], grads64), (key, "fp64", relF(got, grads64[key]), tol)
    # the public surface: predict() on covariances handed in by the caller
    est = uglad_amd.uGLAD_GL()
    est.model_glad, est._fit_cfg = model, dict(L=L, INIT_DIAG=int(g["INIT_DIAG"]), eval_offset=0.1, sqrt_mode=None)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        out = est.predict(S=g["S"])
    warned = [w for w in caught if issubclass(w.category, uglad_amd.UgladRegimeWarning)]
    assert (len(warned) > 0) == (cond > bound), (cond, bound, [str(w.message) for w in caught])
    assert abs(est.predict_cond_max_ - cond) < 1e-3 * cond
    assert np.array_equal(out if out.ndim == 3 else out[None], th)  # (inference path == training forward, bit for bit)


def _cell_dim(name):
    import re

    return int(re.search(r"_d(\d+)_", name).group(1))


@pytest.mark.parametrize("name", [c for c in CELLS if 128 < _cell_dim(c) <= 256])
def test_one_and_many_workgroups_per_matrix_agree(lib, name):
    """D > 128: the cell as one workgroup per matrix and as many workgroups per matrix (csrc/wide_bwd.h, what small batches of large
    matrices run) are the same function up to the summation order of the products: both within the Theta tolerance of the
    reference, and within the gradient noise of each other."""
    import uglad_amd

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    S = torch.from_numpy(g["S"]).cuda()
    out = {}
    for wide in (0, 1):
        lib.set_wide_mode(wide)
        try:
            model = load_model(g)
            with spectral_path(lib):
                theta, loss = uglad_amd.forward_uGLAD(S, model, L=int(g["L"]), INIT_DIAG=int(g["INIT_DIAG"]))
                loss.backward()
        finally:
            lib.set_wide_mode(-1)
        assert max_relF(theta.detach().cpu().numpy(), g["theta_L"]) < TOL
        assert torch.equal(theta, theta.transpose(1, 2))
        out[wide] = (theta.detach().cpu().numpy(), {k: p.grad.cpu().numpy() for k, p in model.named_parameters()}, loss.item())
    assert max_relF(out[1][0], out[0][0]) < 2e-5
    assert abs(out[1][2] - out[0][2]) < 1e-5 * max(1.0, abs(out[0][2]))
    for key in ex.PARAM_KEYS:
        # (the two paths round the forward differently; the gradients amplify that as they do against the reference: DESIGN.md section 2)
        assert relF(out[1][1][key], out[0][1][key]) < 2 * grad_tolerance(name, key), key


def fp64_gradient_sensitivity(S, p64, L, keys, g0, draws=4, eps=2e-7):
    """How far the fp64 oracle's own 42 gradients move (relative Frobenius over the concatenated vector) when S is perturbed at fp32 level:
    the conditioning of the gradient at this input.  O(1e-6) normally; 1e-4 ... 1e-3 where an entry of theta_half sits on its threshold."""
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(draws):
        E = rng.standard_normal(S.shape)
        Sp = S.astype(np.float64) * (1.0 + eps * 0.5 * (E + E.transpose(0, 2, 1)))
        _, tr = ex.glad_forward(Sp, p64, L, 0, mode="ns10")
        g = ex.glad_backward(Sp, p64, L, tr, 0, mode="ns10")
        worst = max(worst, relF(np.concatenate([np.asarray(g[k], np.float64).reshape(-1) for k in keys]), g0))
    return worst


@pytest.mark.parametrize("D,M,N", [(130, 3, 500), (160, 1, 2048), (161, 2, 500), (192, 1, 2048), (193, 3, 500), (224, 2, 500), (224, 2, 2048),
                                   (255, 1, 500), (255, 1, 2048)])
def test_many_workgroups_per_matrix_ragged_sizes(lib, D, M, N):
    """Sizes around the tile and padding edges of the many-workgroup kernels (64 x 64 tiles, halves of 128 columns, 16-column strips):
    both kernel shapes give the same Theta (to the rounding of a differently ordered sum) and gradients within their noise.
    N = 500 is the default conditioning of the synthetic inputs (the generator the bench uses); N = 2048 is better conditioned.  Round 2
    moved this test from N = 500 to N = 2048 without recording why; scripts/wide_ragged_probe.py re-ran both (profiles/r03_wide_ragged_probe_*):
    at (D, N) = (255, 500) with the trained parameters Theta_L is indefinite and the loss is NaN -- in BOTH kernel shapes and in the fp64 oracle
    (min eig(S + tI) = 7e-4): a property of input and parameters, not of the many-workgroup path, which the isfinite assertion tripped over.
    That case is kept here, asserting the NaN on both paths."""
    import uglad_amd
    from uglad_amd.utils.prepare_data import synthetic_covariance_batch

    S = torch.from_numpy(synthetic_covariance_batch(M, D, N, seed=1000 + D)).cuda()
    out = {}
    for wide in (0, 1):
        lib.set_wide_mode(wide)
        try:
            model = trained_model()
            with spectral_path(lib):
                theta, loss = uglad_amd.forward_uGLAD(S, model, L=6)
                loss.backward()
        finally:
            lib.set_wide_mode(-1)
        assert torch.isfinite(theta).all() and torch.equal(theta, theta.transpose(1, 2))
        out[wide] = (theta.detach().cpu().numpy(), torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu().numpy(), loss.item())
    assert max_relF(out[1][0], out[0][0]) < 1e-5  # (measured <= 2.1e-6)
    if (D, N) == (255, 500):  # torch.logdet's NaN for an indefinite Theta_L, identically on both paths (and in the fp64 oracle)
        assert np.isnan(out[0][2]) and np.isnan(out[1][2])
        return
    assert np.isfinite(out[0][2]) and np.isfinite(out[1][2])
    # gradients: the two shapes round the forward differently, and neither is the yardstick -- each is held to the fp64 evaluation of the
    # function (round 3 compared them with each other and granted (224, 500) 1e-3 without saying which was closer; round 4's reordered
    # reduction of the spectrum moved (224, 2048) from 8e-6 to 4.4e-4 between the shapes).  profiles/r04_wide_ragged_vs_fp64.txt
    pz = np.load(os.path.join(GOLDEN, "params_trained.npz"))
    p64 = ex.params64(pz, "")
    ref64, tr = ex.glad_forward(S.cpu().numpy(), p64, 6, 0, mode="ns10")
    g64 = ex.glad_backward(S.cpu().numpy(), p64, 6, tr, 0, mode="ns10")
    v64 = np.concatenate([np.asarray(g64[k], np.float64).reshape(-1) for k, _ in trained_model().named_parameters()])
    e = [relF(out[w][1], v64) for w in (0, 1)]
    et = [max_relF(out[w][0], ref64) for w in (0, 1)]
    print(f"D={D} M={M} N={N}: gradients vs fp64 oracle: one workgroup per matrix {e[0]:.2e}, many workgroups {e[1]:.2e}; "
          f"between the shapes {relF(out[1][1], out[0][1]):.2e}; Theta vs fp64 {et[0]:.2e} / {et[1]:.2e}")
    assert max(et) < 5e-6
    tol = 1e-4
    if max(e) >= tol:
        # Beyond the contract: is it the kernels, or is the gradient ill-determined AT THIS INPUT?  The soft threshold makes dL/dparams discontinuous
        # where |theta_half_ij| crosses rho_ij; the fp64 oracle itself says how close this input is to such a point: its own gradient under
        # fp32-level relative perturbations of S (2e-7, symmetric).  (224, 2, 500): 3e-4 ... 1.3e-3 while Theta moves 1.4e-6; (224, 2, 2048): 3e-4 in
        # one of four draws; every other case of this test: ~1e-6.  The kernels are held to four times that -- profiles/r04_wide_ragged_vs_fp64.txt.
        sens = fp64_gradient_sensitivity(S.cpu().numpy(), p64, 6, [k for k, _ in trained_model().named_parameters()], v64)
        tol = max(tol, 4.0 * sens)
        print(f"    the fp64 oracle's own gradient moves by {sens:.2e} under 2e-7 perturbations of S: tolerance {tol:.2e}")
        assert (D, N) in ((224, 500), (224, 2048)), "a new ill-conditioned case: look at it before accepting it"
    assert max(e) < tol, (e, tol)
    assert abs(out[1][2] - out[0][2]) < 1e-5 * max(1.0, abs(out[0][2]))


def test_intermediates_and_lambdas(lib):
    from uglad_amd.glad import glad as gmod

    g = np.load(os.path.join(GOLDEN, "cell_d64_b4_L30_trained.npz"))
    model = load_model(g)
    with torch.no_grad():
        theta, lam = gmod.glad(torch.from_numpy(g["S"]).cuda(), model, L=30, return_lambdas=True)
    np.testing.assert_allclose(lam.cpu().numpy(), g["lambdas"], rtol=5e-5)
    assert max_relF(theta.cpu().numpy(), g["theta_L"]) < TOL


def test_exact_mode_matches_fp64_oracle(lib):
    from uglad_amd.glad import glad as gmod

    g = np.load(os.path.join(GOLDEN, "cell_d128_b2_L30_trained.npz"))
    model = load_model(g)
    with torch.no_grad():
        th = gmod.glad(torch.from_numpy(g["S"]).cuda(), model, L=30, sqrt_mode="exact")
    ref, _ = ex.glad_forward(g["S"], ex.params64(g, "param."), 30, 0, mode="exact")
    assert max_relF(th.cpu().numpy(), ref) < TOL
    assert max_relF(th.cpu().numpy(), g["theta_L"]) > TOL  # ...and is NOT what the reference computes there


def test_init_diag_and_single_matrix_input(lib):
    from uglad_amd.glad import glad as gmod

    g = np.load(os.path.join(GOLDEN, "cell_d16_b3_L6_diag1_fresh.npz"))
    model = load_model(g)
    with torch.no_grad():
        th = gmod.glad(torch.from_numpy(g["S"][0]).cuda(), model, L=6, INIT_DIAG=1)  # 2-D input -> (1, D, D)
    assert th.shape == (1, 16, 16)


def test_nan_propagates_not_aborts(lib):
    """A non-SPD Theta must give a NaN loss (torch.logdet semantics), not an exception."""
    import uglad_amd

    th = -torch.eye(8, device="cuda")[None].contiguous()
    th[0, 0, 0] = 1.0  # det < 0
    assert torch.isnan(uglad_amd.loss_uGLAD(th, torch.eye(8, device="cuda")[None]))


def test_consensus(lib):
    import uglad_amd

    g = np.load(os.path.join(GOLDEN, "consensus.npz"))
    out = uglad_amd.get_final_precision_from_batch(torch.from_numpy(g["theta_K"]).cuda(), type="min")
    np.testing.assert_array_equal(out.cpu().numpy(), g["out_min"])


# ----------------------------------------------------------------------------------------------- fit-level
def _patched_fit(monkeypatch, g, n_inits):
    from uglad_amd import main

    losses, state = [], {"i": 0}
    real_init, real_fwd = main.init_uGLAD, main.forward_uGLAD

    def init(*a, **k):
        m, _ = real_init(*a, **k)
        i = min(state["i"], n_inits - 1)
        state["i"] += 1
        m.load_state_dict({key: torch.from_numpy(np.array(g[f"init{i}." + key])) for key in ex.PARAM_KEYS})
        return m, main.glad.get_optimizers(m, lr_glad=k.get("lr", a[0] if a else 0.002))

    def fwd(*a, **k):
        th, ls = real_fwd(*a, **k)
        losses.append(float(ls.item()))
        return th, ls

    monkeypatch.setattr(main, "init_uGLAD", init)
    monkeypatch.setattr(main, "forward_uGLAD", fwd)
    return losses


def test_fit_direct_matches_reference_trajectory(lib, monkeypatch):
    import uglad_amd

    g = np.load(os.path.join(GOLDEN, "fit_direct_d25.npz"))
    losses = _patched_fit(monkeypatch, g, 1)
    est = uglad_amd.uGLAD_GL()
    est.fit(g["X"].copy(), epochs=int(g["epochs"]), lr=float(g["lr"]), L=int(g["L"]), verbose=False, mode="direct")
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-4, atol=2e-4)
    err = relF(est.precision_, g["precision_"])
    print(f"fit(direct) 120 epochs: precision_ rel-Frobenius vs reference {err:.2e}")
    assert err < FIT_TOL
    np.testing.assert_allclose(est.covariance_, g["covariance_"], rtol=1e-9, atol=1e-12)
    for key in ex.PARAM_KEYS:
        np.testing.assert_allclose(est.model_glad.state_dict()[key].cpu().numpy(), g["final." + key], rtol=5e-3, atol=5e-4)


def test_fit_direct_beyond_the_eigensolver_matches_reference(lib, monkeypatch):
    """uGLAD_GL.fit(mode="direct") on a 400 x 288 table (D > 256: host covariance, the matrix-iteration cell, L D L^T for Theta_0 and the
    loss), ten epochs against the reference's trajectory."""
    import uglad_amd

    g = np.load(os.path.join(GOLDEN, "fit_direct_d288.npz"))
    losses = _patched_fit(monkeypatch, g, 1)
    est = uglad_amd.uGLAD_GL()
    est.fit(g["X"].copy(), epochs=int(g["epochs"]), lr=float(g["lr"]), L=int(g["L"]), verbose=False, mode="direct")
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-5, atol=0)
    err = relF(est.precision_, g["precision_"])
    print(f"fit(direct) D = 288, 10 epochs: precision_ rel-Frobenius vs reference {err:.2e}")
    assert err < FIT_TOL
    for key in ex.PARAM_KEYS:
        np.testing.assert_allclose(est.model_glad.state_dict()[key].cpu().numpy(), g["final." + key], rtol=1e-4, atol=1e-5)


def test_fit_multitask_matches_reference(lib, monkeypatch):
    import uglad_amd

    g = np.load(os.path.join(GOLDEN, "fit_multitask_d20_k3.npz"))
    losses = _patched_fit(monkeypatch, g, 1)
    est = uglad_amd.uGLAD_multitask()
    est.fit([g[f"X{i}"].copy() for i in range(int(g["n_tasks"]))], epochs=int(g["epochs"]), lr=float(g["lr"]),
            L=int(g["L"]), verbose=False)
    np.testing.assert_allclose(losses, g["losses"], rtol=3e-4, atol=3e-4)
    assert est.precision_.shape == (3, 20, 20)
    err = max_relF(est.precision_, g["precision_"])
    print(f"fit(multitask): precision_ rel-Frobenius vs reference {err:.2e}")
    assert err < FIT_TOL
    np.testing.assert_allclose(est.covariance_, g["covariance_"], rtol=1e-9, atol=1e-12)


def test_fit_missing_matches_reference(lib, monkeypatch):
    import uglad_amd

    g = np.load(os.path.join(GOLDEN, "fit_missing_d20.npz"))
    losses = _patched_fit(monkeypatch, g, 1)
    est = uglad_amd.uGLAD_GL()
    est.fit(g["X"].copy(), epochs=int(g["epochs"]), lr=float(g["lr"]), L=int(g["L"]), verbose=False,
            k_fold=int(g["k_fold"]), mode="missing")
    np.testing.assert_allclose(losses, g["losses"], rtol=3e-4, atol=3e-4)
    err = relF(est.precision_, g["precision_"])
    print(f"fit(missing): precision_ rel-Frobenius vs reference {err:.2e}")
    assert err < FIT_TOL
    np.testing.assert_allclose(est.covariance_, g["covariance_"], rtol=1e-9, atol=1e-12)


def test_fit_cv_matches_reference(lib, monkeypatch):
    import uglad_amd

    g = np.load(os.path.join(GOLDEN, "fit_cv_d16.npz"))
    losses = _patched_fit(monkeypatch, g, int(g["n_inits"]))
    est = uglad_amd.uGLAD_GL()
    est.fit(g["X"].copy(), epochs=int(g["epochs"]), lr=float(g["lr"]), L=int(g["L"]), verbose=False,
            k_fold=int(g["k_fold"]), mode="cv")
    np.testing.assert_allclose(losses, g["losses"], rtol=3e-4, atol=3e-4)
    err = relF(est.precision_, g["precision_"])
    print(f"fit(cv): precision_ rel-Frobenius vs reference {err:.2e}")
    assert err < FIT_TOL


def test_fit_cv_batched_folds_match_reference_golden(lib, monkeypatch):
    """SURVEY 8f N2 against the reference itself: all folds of CV mode as ONE grouped batch (per-fold parameters and lambda
    inside the kernels, best fold picked on the device) must give the estimator the reference's sequential
    run_uGLAD_CV (main.py:428-550) gives from the same initial parameters."""
    import uglad_amd

    g = np.load(os.path.join(GOLDEN, "fit_cv_d16.npz"))
    _patched_fit(monkeypatch, g, int(g["n_inits"]))
    est = uglad_amd.uGLAD_GL()
    est.fit(g["X"].copy(), epochs=int(g["epochs"]), lr=float(g["lr"]), L=int(g["L"]), verbose=False,
            k_fold=int(g["k_fold"]), mode="cv", batched_folds=True)
    err = relF(est.precision_, g["precision_"])
    print(f"fit(cv, batched_folds): precision_ rel-Frobenius vs reference {err:.2e}")
    assert err < FIT_TOL
    for key in ex.PARAM_KEYS:  # the best fold's model after the reference's number of Adam steps
        np.testing.assert_allclose(est.model_glad.state_dict()[key].cpu().numpy(), g["final." + key], rtol=5e-3, atol=5e-4)


def test_fit_direct_converged_matches_reference(lib, monkeypatch):
    """SURVEY.md section 7 hard part 2 / 8d: end-to-end precision_ parity asserted AT CONVERGENCE (600 epochs at D=25, past the
    phase transition around epoch 330 where off-diagonals first survive the threshold): <= 1e-4 relative Frobenius."""
    import uglad_amd

    g = np.load(os.path.join(GOLDEN, "fit_direct_d25_converged.npz"))
    losses = _patched_fit(monkeypatch, g, 1)
    est = uglad_amd.uGLAD_GL()
    est.fit(g["X"].copy(), epochs=int(g["epochs"]), lr=float(g["lr"]), L=int(g["L"]), verbose=False, mode="direct")
    ref = g["losses"]
    dev = np.abs(np.array(losses) - ref) / np.maximum(1.0, np.abs(ref))
    err = relF(est.precision_, g["precision_"])
    print(f"fit(direct) 600 epochs: precision_ rel-Frobenius vs reference {err:.2e}; loss trajectory worst rel dev {dev.max():.2e} "
          f"at epoch {int(dev.argmax())}, final {dev[-1]:.2e}; support mismatches "
          f"{int(np.count_nonzero((est.precision_ != 0) != (g['precision_'] != 0)))}")
    assert err < 1e-4
    assert dev[-50:].max() < 1e-5 and dev.max() < 1e-3
    for key in ex.PARAM_KEYS:
        np.testing.assert_allclose(est.model_glad.state_dict()[key].cpu().numpy(), g["final." + key], rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("golden", ["cell_d25_b1_L15_trained", "cell_d129_b2_L30_trained", "cell_d288_b2_L6_fresh"])
def test_a_whole_pass_can_be_captured_into_the_callers_graph(lib, golden, request):
    """The C entry points neither allocate nor synchronise and keep no state (include/uglad_hip.h), so a caller may capture a whole
    forward + backward pass into a hipGraph of its own and replay it: same bits as plain launches, on every replay.  D = 129: the
    many-workgroup launches of a pass over few large matrices (a dozen launches per cell) inside the capture; D = 288: the matrix-iteration
    path (some fifty launches per cell).  (The library's own
    graph cache was removed in round 3: profiles/r03_fit_small_graph_probe.txt.)"""
    from uglad_amd import _lib

    g = np.load(os.path.join(GOLDEN, golden + ".npz"))
    S = torch.from_numpy(g["S"]).cuda()
    M, D, _ = S.shape
    L = int(g["L"])
    pk = load_model(g).packed().detach().contiguous()
    f32 = dict(dtype=torch.float32, device="cuda")
    Z, half, U = torch.empty(L + 1, M, D, D, **f32), torch.empty(L, M, D, D, **f32), torch.empty(L, M, D, D, **f32)
    beta, lam, lam_in = torch.empty(L, M, D, **f32), torch.empty(L + 1, **f32), torch.empty(L + 1, 2, **f32)
    nfp, nfs, cond, wsp = torch.empty(M, **f32), torch.empty(1, **f32), torch.empty(M, **f32), lib.workspace(M, D, S)
    GL = torch.randn(M, D, D, generator=torch.Generator(device="cuda").manual_seed(3), **f32)
    GL = (GL + GL.transpose(1, 2)).contiguous()
    gb0, gb1 = torch.empty(M, D, D, **f32), torch.empty(M, D, D, **f32)
    grp, glp, gtp, grad = torch.empty(M, 28, **f32), torch.empty(L, M, **f32), torch.empty(M, **f32), torch.empty(42, **f32)
    mode = _lib.SQRT_MODES["ns10"]

    def one_pass():
        lib.glad_forward(S, pk, 1.0, 0, L, Z, half, U, beta, lam, lam_in, nfp, nfs, wsp, mode, cond_max=cond)
        lib.glad_backward(GL, S, pk, 0, L, Z, half, U, beta, lam, lam_in, gb0, gb1, grp, glp, gtp, grad, wsp, mode)

    if D == 129:  # the many-workgroup SPECTRAL kernels are what this case is about (left alone the library takes two 129 x 129 matrices to the
        lib.set_matrix_iteration(0)  # matrix-iteration path, which the D = 288 case covers)
        request.addfinalizer(lambda: lib.set_matrix_iteration(-1))
        wsp = lib.workspace(M, D, S)

    def snapshot():
        torch.cuda.synchronize()
        return Z[L].clone(), grad.clone(), cond.clone(), lam.clone()

    one_pass()
    plain = snapshot()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        one_pass()  # warm-up on the capture stream
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            one_pass()
    for _ in range(3):
        for t in (Z, grad, cond, lam):
            t.zero_()
        graph.replay()
        got = snapshot()
        for name, a, b in zip(("Theta_L", "grad", "cond", "lam"), got, plain):
            assert torch.equal(a, b), (name, int(torch.isnan(a).sum()), int(torch.isnan(b).sum()), a.flatten()[:8], b.flatten()[:8])


def test_fit_cv_batched_folds_match_sequential(lib):
    """the folds of CV mode as ONE grouped batch (per-fold parameters and lambda inside the kernels)."""
    import time

    import uglad_amd

    X = np.random.default_rng(11).standard_normal((240, 20))
    out, secs = [], []
    for batched in (False, True):
        torch.manual_seed(5)
        est = uglad_amd.uGLAD_GL()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        est.fit(X.copy(), epochs=30, lr=0.002, L=15, verbose=False, k_fold=4, mode="cv", batched_folds=batched)
        torch.cuda.synchronize()
        secs.append(time.perf_counter() - t0)
        out.append((est.precision_.copy(), torch.cat([v.detach().cpu().reshape(-1) for v in est.model_glad.state_dict().values()])))
    print(f"CV mode, 4 folds x 30 epochs, D=20: sequential {secs[0]:.3f} s, batched folds {secs[1]:.3f} s")
    assert relF(out[1][0], out[0][0]) < 1e-5
    assert torch.allclose(out[0][1], out[1][1], rtol=0, atol=1e-6)


def test_grouped_pass_equals_independent_passes_gpu(lib):
    import uglad_amd
    from uglad_amd.glad.glad import glad_grouped
    from uglad_amd.utils.prepare_data import synthetic_covariance_batch

    D, L, G, gs = 40, 10, 4, 3
    S = torch.from_numpy(synthetic_covariance_batch(G * gs, D, seed=21)).cuda()
    models = []
    for g in range(G):
        torch.manual_seed(100 + g)
        models.append(uglad_amd.GladParams(1.0 + 0.1 * g, device="cuda"))
    P = torch.stack([m.packed().detach() for m in models]).requires_grad_(True)
    W = torch.randn(G * gs, D, D, device="cuda")
    th = glad_grouped(S, P, L=L)
    (th * W).sum().backward()
    for g in range(G):
        sl = slice(g * gs, (g + 1) * gs)
        t1 = uglad_amd.glad(S[sl], models[g], L=L)
        (t1 * W[sl]).sum().backward()
        assert torch.equal(t1.detach(), th[sl].detach())
        g1 = torch.cat([p.grad.reshape(-1) for p in models[g].parameters()])
        assert torch.equal(g1, P.grad[g])


def test_fit_cv_parallel_folds_is_bit_identical_to_sequential(lib):
    """the folds of CV mode on separate host threads / HIP streams."""
    import time

    import uglad_amd

    X = np.random.default_rng(11).standard_normal((240, 20))
    out, secs = [], []
    for par in (False, True):
        torch.manual_seed(5)
        est = uglad_amd.uGLAD_GL()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        est.fit(X.copy(), epochs=30, lr=0.002, L=15, verbose=False, k_fold=4, mode="cv", parallel_folds=par)
        torch.cuda.synchronize()
        secs.append(time.perf_counter() - t0)
        out.append((est.precision_.copy(), [v.detach().cpu().clone() for v in est.model_glad.state_dict().values()]))
    print(f"CV mode, 4 folds x 30 epochs, D=20: sequential {secs[0]:.3f} s, parallel folds {secs[1]:.3f} s")
    assert np.array_equal(out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        assert torch.equal(a, b)


def test_predict_and_errors(lib):
    import uglad_amd

    X = np.random.default_rng(3).standard_normal((200, 12))
    est = uglad_amd.uGLAD_GL()
    with pytest.raises(ValueError):
        est.predict(X)
    with pytest.raises(ValueError):
        est.fit(X, epochs=2, verbose=False, mode="bogus")
    est.fit(X, epochs=3, L=5, verbose=False)  # epochs < 10 works here (the reference divides by zero)
    p = est.predict(X)  # trained parameters AFTER the last Adam step; precision_ is the last training forward
    assert p.shape == (12, 12) and np.isfinite(p).all() and np.array_equal(p, p.T)
    assert relF(p, est.precision_) < 0.05
    pm = est.predict(S=np.stack([est.covariance_, est.covariance_]))
    assert pm.shape == (2, 12, 12) and np.array_equal(pm[0], pm[1])


# ----------------------------------------------------------------------------------------------- config 2 / 3 sized
@pytest.mark.parametrize("D,M,Mcpu", [(64, 128, 8), (128, 1024, 4)])
def test_full_size_properties_and_subsample_parity(lib, D, M, Mcpu):
    """BASELINE configs 2 and 3 at full size: the oracle (NS-faithful CPU restatement, itself golden-pinned) checks a
    subsample; properties that need no oracle cover the rest."""
    import uglad_amd
    from uglad_amd.glad import glad as gmod
    from uglad_amd.utils.prepare_data import synthetic_covariance_batch

    L = 30
    # the first 64 matrices ARE bench.py's inputs (its generator and seed: task i = default_rng(1234 + i)) -- so the sub-batch checked against
    # the oracle below is the head of the benchmark's own batch; the rest are convex mixtures of them (cheap, stay in uGLAD's input regime:
    # sampling all 1024 takes half a minute on one core)
    NB = 64
    base = synthetic_covariance_batch(NB, D, seed=1234)
    rng = np.random.default_rng(D)
    w = rng.dirichlet(np.ones(NB) * 0.5, size=M).astype(np.float32)
    w[:NB] = np.eye(NB, dtype=np.float32)
    S = torch.from_numpy(np.einsum("mk,kij->mij", w, base)).cuda().contiguous()
    model = trained_model()
    theta, loss = uglad_amd.forward_uGLAD(S, model, L=L)
    loss.backward()
    g1 = model.packed().detach().clone(), torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    assert torch.isfinite(theta).all() and torch.isfinite(loss) and torch.isfinite(g1[1]).all()
    assert torch.equal(theta, theta.transpose(1, 2))
    # determinism: a second pass is bit-identical (no atomics anywhere on the path)
    model.zero_grad()
    theta2, loss2 = uglad_amd.forward_uGLAD(S, model, L=L)
    loss2.backward()
    g2 = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert torch.equal(theta, theta2) and torch.equal(loss, loss2) and torch.equal(g1[1], g2)
    # inference path (ping-pong buffers, nothing saved) == training forward
    with torch.no_grad():
        theta3, lam3 = gmod.glad(S, model, L=L, return_lambdas=True)
    assert torch.equal(theta, theta3)
    # subsample parity: run the first Mcpu matrices ALONE on the GPU and through the CPU oracle.  lambda_k depends on the
    # batch mean, so the sub-batch is its own problem on both sides.
    Ssub = S[:Mcpu].contiguous()
    th_gpu, loss_gpu = uglad_amd.forward_uGLAD(Ssub, model, L=L)
    model.zero_grad()
    loss_gpu.backward()
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    th_cpu, loss_cpu = ns.forward_uGLAD(Ssub.cpu(), p, L=L)
    loss_cpu.backward()
    err = max_relF(th_gpu.detach().cpu().numpy(), th_cpu.detach().numpy())
    print(f"D={D} M={M}: sub-batch of {Mcpu} vs NS-faithful CPU oracle: Theta rel-Frobenius {err:.2e}; "
          f"loss {loss_gpu.item():.5f} vs {loss_cpu.item():.5f}")
    assert err < TOL
    assert abs(loss_gpu.item() - loss_cpu.item()) < 1e-4 * abs(loss_cpu.item())
    sd = dict(model.named_parameters())
    gerr = {key: relF(sd[key].grad.cpu().numpy(), p[key].grad.numpy()) for key in ex.PARAM_KEYS}
    worst = max(gerr, key=gerr.get)
    print(f"D={D} M={M}: sub-batch gradients vs NS-faithful CPU oracle (fp32): worst {worst} {gerr[worst]:.2e}")
    for key in ex.PARAM_KEYS:  # the gradient contract; the fp32 oracle's own noise at these sizes is 2e-5 ... 5e-5 (grad_noise_floor.json)
        ref, got = p[key].grad.numpy(), sd[key].grad.cpu().numpy()
        assert gerr[key] < GRAD_CONTRACT or tiny_gradient(got, ref, {k: p[k].grad.numpy() for k in ex.PARAM_KEYS}), (key, gerr[key], got, ref)


def test_config5_shape_missing_data_consensus(lib):
    """BASELINE config 5 shape: K=8 sub-sample covariances of order D=256 (beyond the LDS-resident size: the kernels run on
    workspace slabs), loss against ONE full covariance (divisor 1), consensus at the end.  A K=2 sub-batch is checked against
    the NS-faithful CPU oracle; the full K=8 pass is checked for determinism, symmetry and the consensus identity."""
    import uglad_amd
    from uglad_amd.utils.prepare_data import get_covariance, get_data

    D, K, L = 256, 8, 30
    X, _ = get_data(D, (0.1, 0.2), 1024, 1, eig_offset=1.0, rng=55)
    X = X[0]
    X = (X - X.min(0)) / (X.max(0) - X.min(0))
    S_full = torch.from_numpy(get_covariance([X])[0].astype(np.float32))[None].cuda()
    folds = np.array_split(np.arange(X.shape[0]), K)
    S_K = torch.from_numpy(np.stack([get_covariance([np.delete(X, f, axis=0)])[0] for f in folds]).astype(np.float32)).cuda()
    model = load_model(np.load(os.path.join(GOLDEN, "params_fresh.npz")), "")
    theta, loss = uglad_amd.forward_uGLAD(S_K, model, L=L, loss_Sb=S_full)
    loss.backward()
    g1 = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    assert torch.isfinite(theta).all() and torch.isfinite(loss) and torch.isfinite(g1).all()
    assert torch.equal(theta, theta.transpose(1, 2))
    model.zero_grad()
    theta2, loss2 = uglad_amd.forward_uGLAD(S_K, model, L=L, loss_Sb=S_full)
    loss2.backward()
    assert torch.equal(theta, theta2) and torch.equal(loss, loss2)
    assert torch.equal(g1, torch.cat([p.grad.reshape(-1) for p in model.parameters()]))
    cons = uglad_amd.get_final_precision_from_batch(theta.detach(), type="min")
    ref = ns.consensus_min(theta.detach().cpu())
    assert torch.equal(cons.cpu(), ref)
    # K=2 sub-batch against the oracle
    th_g, ls_g = uglad_amd.forward_uGLAD(S_K[:2].contiguous(), model, L=L, loss_Sb=S_full)
    p = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    with torch.no_grad():
        th_c, ls_c = ns.forward_uGLAD(S_K[:2].cpu(), p, L=L, loss_Sb=S_full.cpu())
    err = max_relF(th_g.detach().cpu().numpy(), th_c.numpy())
    print(f"config-5 shape (D=256): K=2 sub-batch vs NS-faithful CPU oracle: Theta rel-Frobenius {err:.2e}; "
          f"loss {ls_g.item():.5f} vs {ls_c.item():.5f}")
    assert err < TOL
    assert abs(ls_g.item() - ls_c.item()) < 1e-4 * abs(ls_c.item())


def test_sharded_equals_unsharded_in_process(lib):
    """Two 'ranks' in one process through an injected collective that sums the shard partials: Theta and the 42 gradients of
    the two half-batches must equal the unsharded run (exchange sites i and ii of uglad_amd/dist.py)."""
    import uglad_amd
    from uglad_amd.dist import Collective
    from uglad_amd.utils.prepare_data import synthetic_covariance_batch

    S = torch.from_numpy(synthetic_covariance_batch(6, 32, seed=99)).cuda()
    model = trained_model()
    L = 8
    theta, loss = uglad_amd.forward_uGLAD(S, model, L=L)
    loss.backward()
    gfull = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    lam_full = None

    # replay: rank r needs the OTHER shard's per-step partial sums; record them from a first pass of each shard driven by
    # the full-batch sums (a fake all-reduce that returns the known global value)
    from uglad_amd.glad import glad as gmod
    with torch.no_grad():
        _, lam_full = gmod.glad(S, model, L=L, return_lambdas=True)

    class Replay(Collective):
        world_size = 2

        def __init__(self, sums):
            self.sums, self.k = sums, 0

        def all_reduce_sum(self, t):
            t.copy_(self.sums[self.k])
            self.k += 1
            return t

    # global per-step sums from the unsharded run: n_k * M = lam_in[k+1][0] * M ; recompute them from the two shards
    class Record(Collective):
        def __init__(self):
            self.vals = []

        def all_reduce_sum(self, t):
            self.vals.append(t.clone())
            return t

    rec = Record()
    with torch.no_grad():
        gmod.glad(S, model, L=L, collective=rec, global_batch=6)
    grads, thetas = [], []
    for lo, hi in ((0, 3), (3, 6)):
        model.zero_grad()
        th, ls = uglad_amd.forward_uGLAD(S[lo:hi].contiguous(), model, L=L, collective=Replay(rec.vals), global_batch=6)
        ls.backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone())
        thetas.append(th.detach())
    assert torch.allclose(torch.cat(thetas), theta.detach(), rtol=0, atol=0)
    assert torch.allclose(grads[0] + grads[1], gfull, rtol=2e-5, atol=1e-6)


def test_sharded_whole_pass_entry_with_rccl_and_with_an_injected_exchange(lib):
    """uglad_glad_forward_sharded: one rank's whole pass from ONE C call, the per-step SUM through a function the caller hands in.
      (a) RCCL -- a communicator of the library's own (uglad_rccl_comm_init, world of one rank: all this box offers) whose ncclAllReduce the
          pass issues itself on the compute stream: the sum over one rank is the identity, so the pass must equal uglad_glad_forward bit for
          bit; this executes the dlopen / ncclCommInitRank / ncclAllReduce-on-our-stream code that an 8-GPU run takes;
      (b) an injected exchange that adds the OTHER shard's recorded per-step partial sums (two shards of three matrices, global batch six):
          both shards must reproduce the unsharded Theta bit for bit -- the exchange sits between the local sum and LambdaNN, and the
          divisor is the global batch."""
    import ctypes

    from uglad_amd import _lib
    from uglad_amd.dist import Collective
    from uglad_amd.glad import glad as gmod
    from uglad_amd.utils.prepare_data import synthetic_covariance_batch

    S = torch.from_numpy(synthetic_covariance_batch(6, 32, seed=99)).cuda()
    model = trained_model()
    pk = model.packed().detach().contiguous()
    L, D, mode = 8, 32, _lib.SQRT_MODES["ns10"]
    f32 = dict(dtype=torch.float32, device="cuda")

    def buffers(M):
        return dict(Z=torch.empty(L + 1, M, D, D, **f32), half=torch.empty(L, M, D, D, **f32), U=torch.empty(L, M, D, D, **f32),
                    beta=torch.empty(L, M, D, **f32), lam=torch.empty(L + 1, **f32), lam_in=torch.empty(L + 1, 2, **f32),
                    nfp=torch.empty(M, **f32), nfs=torch.empty(1, **f32))

    def run(Sx, b, sharded=None, m_global=None):
        wsp = lib.workspace(Sx.shape[0], D, Sx)
        args = (Sx, pk, 1.0, 0, L, b["Z"], b["half"], b["U"], b["beta"], b["lam"], b["lam_in"], b["nfp"], b["nfs"], wsp, mode)
        if sharded is None:
            lib.glad_forward(*args)
        else:
            lib.glad_forward_sharded(*args, m_global, sharded)
        torch.cuda.synchronize()

    full = buffers(6)
    run(S, full)
    # (a) RCCL, one rank
    comm = lib.rccl_comm_init(lib.rccl_unique_id(), 1, 0)
    try:
        one = buffers(6)
        run(S, one, sharded=lib.rccl_exchange(comm), m_global=6)
        assert torch.equal(one["Z"], full["Z"]) and torch.equal(one["lam"], full["lam"]) and torch.equal(one["U"], full["U"])
    finally:
        lib.rccl_comm_destroy(comm)

    # (b) two shards, the other shard's per-step sums injected
    class Record(Collective):
        def __init__(self):
            self.vals = []

        def all_reduce_sum(self, t):
            self.vals.append(t.clone())
            return t

    local = []
    for lo, hi in ((0, 3), (3, 6)):  # per-step LOCAL sums of each shard along the global trajectory: drive the steps with the global lambdas
        rec, Zs = Record(), full["Z"]
        nfp, nfs = torch.empty(3, **f32), torch.empty(1, **f32)
        wsp = lib.workspace(3, D, S)
        for k in range(L):
            lib.cell_fwd(S[lo:hi].contiguous(), Zs[k, lo:hi].contiguous(), full["lam"][k:k + 1], pk, torch.empty(3, D, D, **f32), None, None, None,
                         nfp, wsp, mode)
            lib.sum_partials(nfp, nfs)
            rec.all_reduce_sum(nfs)
        local.append(rec.vals)
    for r, (lo, hi) in enumerate(((0, 3), (3, 6))):
        b = buffers(3)
        other, step = local[1 - r], [0]

        def exchange(buf, n, ctx, stream, b=b, other=other, step=step):
            assert n == 1 and buf == b["nfs"].data_ptr()
            b["nfs"].add_(other[step[0]])  # (enqueued on the current stream, like a collective)
            step[0] += 1
            return 0

        cb = _lib.HipLib.ALLREDUCE_FN(exchange)
        run(S[lo:hi].contiguous(), b, sharded=(cb, None), m_global=6)
        assert step[0] == L

        class AddOther(Collective):  # the same exchange through the per-step Python loop (what gloo rehearsals and fakes run)
            world_size, k = 2, 0

            def all_reduce_sum(self, t, other=other):
                t.add_(other[self.k])
                self.k += 1
                return t

        with torch.no_grad():
            th_py, lam_py = gmod.glad(S[lo:hi].contiguous(), model, L=L, collective=AddOther(), global_batch=6, return_lambdas=True)
        assert torch.equal(b["Z"][L], th_py) and torch.equal(b["lam"], lam_py)  # one C call == the per-step loop, bit for bit
        # ... and both are the unsharded pass up to the association of the batch sum ((a0+a1+a2)+(b0+b1+b2) instead of a0+...+b2)
        assert max_relF(b["Z"][L].cpu().numpy(), full["Z"][L, lo:hi].cpu().numpy()) < 1e-6
        assert torch.allclose(b["lam"], full["lam"], rtol=1e-6, atol=0)


@pytest.mark.gpu
def test_direct_mode_nan_break_leaves_the_reference_state(lib, monkeypatch):
    """The reference stops at the first NaN loss before that epoch's backward/step (main.py:401-406); the lagged device-side
    check must leave the same model: parameters as they were before the NaN epoch, predTheta of the NaN epoch."""
    import uglad_amd
    from uglad_amd import main as um

    X = np.random.default_rng(4).standard_normal((120, 10))
    real = um.forward_uGLAD
    calls = {"n": 0}
    NAN_AT = 6

    def flaky(*a, **k):
        theta, loss = real(*a, **k)
        calls["n"] += 1
        if calls["n"] == NAN_AT + 1:  # epochs count from 0
            loss = loss * float("nan")
        return theta, loss

    torch.manual_seed(9)
    ref = uglad_amd.uGLAD_GL()
    ref.fit(X.copy(), epochs=NAN_AT, lr=0.01, L=5, verbose=False)  # NAN_AT clean epochs: the state right before the NaN epoch
    want = [v.detach().cpu().clone() for v in ref.model_glad.state_dict().values()]

    monkeypatch.setattr(um, "forward_uGLAD", flaky)
    torch.manual_seed(9)
    est = uglad_amd.uGLAD_GL()
    est.fit(X.copy(), epochs=40, lr=0.01, L=5, verbose=False)
    got = [v.detach().cpu() for v in est.model_glad.state_dict().values()]
    assert calls["n"] <= NAN_AT + 2  # at most one speculative epoch after the NaN one
    for a, b in zip(want, got):
        assert torch.equal(a, b)


def test_config4_workload_eight_shards_equal_unsharded(lib):
    """BASELINE config 4's workload (M=8192, D=128, L=30) on ONE GPU: the unsharded pass against eight shards of 1024 driven in
    lockstep by eight host threads through an injected collective that really sums the shards' partials (exchange sites i and
    ii of uglad_amd/dist.py; the sum runs in rank order, so every 'rank' sees the same bits).  Only reduction-order noise may
    separate the two: lambda_k differs in its last bits, Theta follows."""
    from uglad_amd.utils.prepare_data import synthetic_covariance_batch

    M, D = 8192, 128
    base = synthetic_covariance_batch(16, D, seed=808)
    w = np.random.default_rng(8).dirichlet(np.ones(16) * 0.5, size=M).astype(np.float32)
    S = torch.from_numpy(np.einsum("mk,kij->mij", w, base)).cuda().contiguous()
    _shards_equal_unsharded(S, L=30, W=8, what="config 4 workload")


def test_config5_partitioning_one_large_matrix_per_shard(lib):
    """BASELINE config 5's partitioning (K=8 tasks of D=256, ONE matrix per GPU) on one GPU: eight shards of one 256 x 256 matrix
    each -- the many-workgroups-per-matrix kernels (csrc/wide_bwd.h, wide_fwd.h) at M = 1 -- against the unsharded batch of eight."""
    from uglad_amd.utils.prepare_data import synthetic_covariance_batch

    S = torch.from_numpy(synthetic_covariance_batch(8, 256, seed=55)).cuda().contiguous()
    with spectral_path(lib):
        _shards_equal_unsharded(S, L=30, W=8, what="config 5 partitioning")


def test_config5_partitioning_on_the_matrix_iteration_path(lib):
    """The same partitioning on the path the library takes BY ITSELF for one 256 x 256 matrix per GPU (csrc/wide_ns.h; the unsharded batch of
    eight is pinned to it as well, the automatic rule would give eight matrices to the eigensolver)."""
    from uglad_amd.utils.prepare_data import synthetic_covariance_batch

    S = torch.from_numpy(synthetic_covariance_batch(8, 256, seed=55)).cuda().contiguous()
    lib.set_matrix_iteration(1)
    try:
        _shards_equal_unsharded(S, L=30, W=8, what="config 5 partitioning, matrix iteration")
    finally:
        lib.set_matrix_iteration(-1)


def _shards_equal_unsharded(S, L, W, what):
    import threading

    import uglad_amd
    from uglad_amd import main as um
    from uglad_amd.dist import Collective

    M = S.shape[0]
    model = trained_model()
    theta, loss = uglad_amd.forward_uGLAD(S, model, L=L)
    loss.backward()
    gfull = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    theta, loss = theta.detach().clone(), loss.detach().clone()
    model.zero_grad()
    torch.cuda.synchronize()

    class Lockstep(Collective):
        """In-process stand-in for the RCCL all-reduce: every rank deposits its tensor, all wait, every rank adds the W
        deposits in rank order.  All ranks enqueue on the same (default) stream, so host-side barriers order the device work."""
        world_size = W
        slots, bar = [None] * W, threading.Barrier(W)

        def __init__(self, rank):
            self.rank = rank

        def all_reduce_sum(self, t):
            Lockstep.slots[self.rank] = t.clone()
            Lockstep.bar.wait()
            acc = Lockstep.slots[0].clone()
            for r in range(1, W):
                acc += Lockstep.slots[r]
            Lockstep.bar.wait()
            t.copy_(acc)
            return t

    shard_theta, shard_grads, shard_loss, errors = [None] * W, [None] * W, [None] * W, []
    params0 = model.packed().detach().clone()

    def rank_main(r):
        try:
            torch.cuda.set_device(S.device)
            m = trained_model()
            coll = Lockstep(r)
            lo, hi = coll.shard(M)
            th, ls = um.forward_uGLAD(S[lo:hi].contiguous(), m, L=L, collective=coll, global_batch=M)
            ls.backward()
            tot = um._allreduce_grads(m, ls, coll)
            shard_theta[r] = th.detach()
            shard_grads[r] = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
            shard_loss[r] = tot.detach().clone()
        except BaseException as exc:  # noqa: BLE001 -- surfaced below; break the barrier so the others do not hang
            errors.append(exc)
            Lockstep.bar.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(W)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    torch.cuda.synchronize()
    assert torch.equal(params0, model.packed().detach())
    th8 = torch.cat(shard_theta)
    errs = ((th8 - theta).flatten(1).norm(dim=1) / theta.flatten(1).norm(dim=1))
    gerr = float((shard_grads[0] - gfull).norm() / gfull.norm())
    print(f"{what}, {W} shards vs unsharded: Theta rel-Frobenius max {errs.max().item():.2e}, "
          f"42 gradients {gerr:.2e}, loss {shard_loss[0].item():.6f} vs {loss.item():.6f}")
    for r in range(1, W):  # every rank holds the same all-reduced gradients and loss, bit for bit
        assert torch.equal(shard_grads[r], shard_grads[0]) and torch.equal(shard_loss[r], shard_loss[0])
    assert errs.max().item() < 1e-5
    assert gerr < 1e-5
    assert abs(shard_loss[0].item() - loss.item()) < 1e-6 * abs(loss.item())


@pytest.mark.parametrize("D", [7, 33, 64, 100, 128])
def test_cholesky_inverse_logdet_and_the_eigen_fallback(lib, D):
    """Theta_0 = (S + tI)^-1 and the loss's logdet / Theta^-1 by blocked Cholesky (csrc/chol.h, D <= 128) against numpy fp64 -- the LU-based
    primitives the reference calls there (glad.py:115, main.py:307) are accurate to ~1e-7 --, with, in the same batch, an INDEFINITE matrix
    (a pivot fails: the eigen path computes it; the inverse exists, torch.logdet gives NaN for det < 0) and a matrix holding a NaN."""
    if os.environ.get("UGLAD_CHOLESKY", "1")[:1] == "0":
        pytest.skip("UGLAD_CHOLESKY=0 (scripts/gpu_toggle_matrix.sh): the eigen path computes every matrix, there are no flags to inspect")
    rng = np.random.default_rng(D)
    M = 6
    A = rng.standard_normal((M, D, 2 * D))
    S = (A @ A.transpose(0, 2, 1) / (2 * D)).astype(np.float32)  # SPD, cond ~ 30
    S[4] = S[4] - 1.5 * np.eye(D, dtype=np.float32)  # indefinite (eigenvalues of both signs)
    St = torch.from_numpy(S).cuda()
    pk = torch.zeros(42, device="cuda")
    pk[0] = 0.05
    th0 = torch.empty_like(St)
    wsp = lib.workspace(M, D, St)
    lib.init_theta(St, pk, 0, th0, wsp)
    flags = wsp[M * 3 * (32 * ((D + 31) // 32)):][:M].view(torch.int32).cpu().numpy()
    assert flags.tolist() == [0, 0, 0, 0, 1, 0]  # only the indefinite matrix went to the eigen path
    ref = np.linalg.inv(S.astype(np.float64) + 0.05 * np.eye(D))
    got = th0.cpu().numpy()
    err = max(relF(got[m], ref[m]) for m in (0, 1, 2, 3, 5))
    assert err < 1e-6, err  # (measured 1e-7 ... 4e-7)
    assert relF(got[4], ref[4]) < 1e-3  # the indefinite one (eigenvalues on both sides of zero: ill-conditioned), by the eigen path
    assert torch.equal(th0, th0.transpose(1, 2))
    # loss: -logdet + tr(S Theta) and Theta^-1; Theta = the SPD matrices, one indefinite with det < 0, one with a NaN
    Th = S.copy()
    Th[4] = S[0]
    Th[4][0, 0] -= 100.0  # one negative eigenvalue: det < 0 -> NaN
    Th[5][1, 2] = Th[5][2, 1] = np.nan
    Tt = torch.from_numpy(Th).cuda()
    lp, tinv = torch.empty(M, device="cuda"), torch.empty_like(Tt)
    lib.loss_fwd(Tt, St[:1].contiguous(), None, lp, tinv, wsp)
    lp = lp.cpu().numpy()
    assert np.isnan(lp[4]) and np.isnan(lp[5])
    for m in range(4):
        sign, ld = np.linalg.slogdet(Th[m].astype(np.float64))
        want = -ld + float(np.sum(S[0].astype(np.float64) * Th[m].astype(np.float64).T))
        assert abs(lp[m] - want) < 2e-6 * max(1.0, abs(want)) + 1e-4, (m, lp[m], want)
    assert max_relF(tinv[:4].cpu().numpy(), np.linalg.inv(Th[:4].astype(np.float64))) < 1e-6
