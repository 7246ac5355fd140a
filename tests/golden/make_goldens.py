#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference on CPU.

Runs only in the build container (needs /root/reference, which never travels to the GPU
box).  Recipe = SURVEY.md Appendix A: PYTHONPATH=/root/reference, a stub for the absent
`pyvis` (only an annotation + viz functions touch it), MPLBACKEND=Agg, scratch CWD because
the reference writes loss_curve.png into the CWD.

    cd /tmp && python /root/repo/tests/golden/make_goldens.py

What is captured (inputs AND expected outputs, never reference source):
  params_{fresh,trained}.npz   the 11 tensors / 42 floats of GladParams.state_dict()
  cell_*.npz                   S, L, INIT_DIAG -> lambda_k, normF_k, Theta_{k+1/2}, Theta_k for selected k,
                               Theta_L, loss, the 11 parameter gradients of loss.backward()
  consensus.npz                get_final_precision_from_batch(type="min")
  fit_*.npz                    X -> init state_dict(s), loss per forward call, precision_, covariance_, location_,
                               final state_dict for uGLAD_GL.fit(direct/cv/missing) and uGLAD_multitask.fit
Intermediates are recorded by wrapping model.eta_forward / model.lambda_forward around the
reference's own glad.glad() call, so every number comes out of the reference's code path.
"""
import contextlib
import io
import os
import sys
import types

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REF = "/root/reference"
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

pv = types.ModuleType("pyvis")
pv.network = types.ModuleType("pyvis.network")
pv.network.Network = object
sys.modules["pyvis"] = pv
sys.modules["pyvis.network"] = pv.network

import numpy as np  # noqa: E402
import torch  # noqa: E402

import uglad.main as uG  # noqa: E402  (the reference)
from uglad.glad import glad as ref_glad  # noqa: E402

from uglad_amd.utils import prepare_data as pd_new  # noqa: E402  (this repo: seeded input generator only)

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def sd_to_np(sd):
    return {k: v.detach().cpu().numpy().astype(np.float32).copy() for k, v in sd.items()}


def load_params(model, sd_np):
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()})


def synth_S(K, D, seed, N=None):
    return pd_new.synthetic_covariance_batch(K, D, N, seed=seed)


def synth_X(D, N, seed, eig_offset=1.0):
    rng = np.random.default_rng(seed)
    Xb, Pb = pd_new.get_data(D, (0.1, 0.2), N, 1, eig_offset=eig_offset, rng=rng)
    return Xb[0], Pb[0]


# ------------------------------------------------------------------ cell-level capture
def capture_cell(name, S, params, L, INIT_DIAG, keep_k, loss_S=None, struct=None, keep_init=True):
    model, _ = uG.init_uGLAD(0.002)
    load_params(model, params)
    rec = {"th_half": [], "th_out": [], "lam_in": [], "lam_out": []}
    eta0, lam0 = model.eta_forward, model.lambda_forward

    def eta(X, Sx, k, F3=None):
        out = eta0(X, Sx, k, F3)
        rec["th_half"].append(X.detach().numpy().copy())
        rec["th_out"].append(out.detach().numpy().copy())
        if k == 0:
            rec["th_init"] = F3.detach().numpy().copy()
        return out

    def lam(normF, prev, k=0):
        out = lam0(normF, prev, k)
        rec["lam_in"].append([float(normF), float(prev)])
        rec["lam_out"].append(float(out))
        return out

    model.eta_forward, model.lambda_forward = eta, lam
    St = torch.from_numpy(S)
    kw = {}
    if loss_S is not None:
        kw["loss_Sb"] = torch.from_numpy(loss_S)
    if struct is not None:
        kw["struct_theta"] = torch.from_numpy(struct)
    with quiet():
        theta, loss = uG.forward_uGLAD(St, model, L=L, INIT_DIAG=INIT_DIAG, **kw)
    loss.backward()
    out = {
        "S": S.astype(np.float32),
        "L": np.int64(L),
        "INIT_DIAG": np.int64(INIT_DIAG),
        "lambdas": np.array(rec["lam_out"], dtype=np.float32),  # lambda_0..lambda_L
        "lambda_inputs": np.array(rec["lam_in"], dtype=np.float32),  # (normF or lambda_init, lambda_prev)
        "theta_L": theta.detach().numpy().astype(np.float32),
        "loss": np.float32(loss.item()),
        "keep_k": np.array(keep_k, dtype=np.int64),
    }
    if keep_init:
        out["theta_init"] = rec["th_init"]
    for k in keep_k:
        out[f"theta_half_{k}"] = rec["th_half"][k]
        out[f"theta_out_{k}"] = rec["th_out"][k]
    if loss_S is not None:
        out["loss_S"] = loss_S.astype(np.float32)
    if struct is not None:
        out["struct"] = struct.astype(np.float32)
    for k, p in model.named_parameters():
        out["grad." + k] = p.grad.detach().numpy().astype(np.float32).copy()
    for k, v in params.items():
        out["param." + k] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss {loss.item():.6f}  lambda_L {rec['lam_out'][-1]:.5f}  "
          f"nnz {np.count_nonzero(out['theta_L'])}/{out['theta_L'].size}")


# ------------------------------------------------------------------ fit-level capture
class FitRecorder:
    """Wraps the reference's init_uGLAD / forward_uGLAD to log what fit() does."""

    def __enter__(self):
        self.inits, self.losses = [], []
        self._init, self._fwd = uG.init_uGLAD, uG.forward_uGLAD

        def init(*a, **k):
            m, o = self._init(*a, **k)
            self.inits.append(sd_to_np(m.state_dict()))
            return m, o

        def fwd(*a, **k):
            th, ls = self._fwd(*a, **k)
            self.losses.append(float(ls.item()))
            return th, ls

        uG.init_uGLAD, uG.forward_uGLAD = init, fwd
        return self

    def __exit__(self, *exc):
        uG.init_uGLAD, uG.forward_uGLAD = self._init, self._fwd


def save_fit(name, obj, rec, extra):
    out = dict(extra)
    out["losses"] = np.array(rec.losses, dtype=np.float64)
    out["n_inits"] = np.int64(len(rec.inits))
    for i, sd in enumerate(rec.inits):
        for k, v in sd.items():
            out[f"init{i}.{k}"] = v
    for k, v in sd_to_np(obj.model_glad.state_dict()).items():
        out["final." + k] = v
    out["precision_"] = np.asarray(obj.precision_)
    out["covariance_"] = np.asarray(obj.covariance_)
    if getattr(obj, "location_", None) is not None:
        out["location_"] = np.asarray(obj.location_)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: {len(rec.losses)} forward calls, loss {rec.losses[0]:.5f} -> {rec.losses[-1]:.5f}")


def main():
    os.chdir("/tmp")
    # ---------------- parameters
    torch.manual_seed(123)
    m, _ = uG.init_uGLAD(0.002)
    fresh = sd_to_np(m.state_dict())
    np.savez(os.path.join(OUT, "params_fresh.npz"), **fresh)

    X25, P25 = synth_X(25, 500, 77)
    torch.manual_seed(123)
    g = uG.uGLAD_GL()
    with quiet():
        g.fit(X25.copy(), centered=False, epochs=500, lr=0.002, INIT_DIAG=0, L=15, verbose=False, mode="direct")
    trained = sd_to_np(g.model_glad.state_dict())
    np.savez(os.path.join(OUT, "params_trained.npz"), **trained)
    print("trained theta_init_offset", trained["theta_init_offset"])

    # ---------------- cells
    S16 = synth_S(3, 16, 500)
    capture_cell("cell_d16_b3_L6_diag0_fresh", S16, fresh, 6, 0, [0, 1, 5])
    capture_cell("cell_d16_b3_L6_diag1_fresh", S16, fresh, 6, 1, [0, 1, 5])
    capture_cell("cell_d16_b3_L6_diag0_trained", S16, trained, 6, 0, [0, 1, 5])
    S25 = synth_S(1, 25, 600)
    capture_cell("cell_d25_b1_L15_fresh", S25, fresh, 15, 0, [0, 1, 14])
    capture_cell("cell_d25_b1_L15_trained", S25, trained, 15, 0, [0, 1, 14])
    S20 = synth_S(5, 20, 650)
    capture_cell("cell_d20_b5_L15_trained", S20, trained, 15, 0, [0, 14])
    S64 = synth_S(4, 64, 700)
    capture_cell("cell_d64_b4_L30_fresh", S64, fresh, 30, 0, [0, 29])
    capture_cell("cell_d64_b4_L30_trained", S64, trained, 30, 0, [0, 29])
    S128 = synth_S(2, 128, 800)
    capture_cell("cell_d128_b2_L30_fresh", S128, fresh, 30, 0, [29], keep_init=False)
    capture_cell("cell_d128_b2_L30_trained", S128, trained, 30, 0, [0, 29], keep_init=False)
    S256 = synth_S(1, 256, 900)
    capture_cell("cell_d256_b1_L30_fresh", S256, fresh, 30, 0, [], keep_init=False)
    # missing-data shaped call: glad on K sub-sample covariances, loss against ONE full covariance (main.py:620-622)
    rng = np.random.default_rng(4242)
    Xm, _ = synth_X(20, 600, 910)
    Xm = (Xm - Xm.min(0)) / (Xm.max(0) - Xm.min(0))
    Sfull = pd_new.get_covariance([Xm])[0].astype(np.float32)[None]
    idx = np.array_split(np.arange(600), 3)
    SK = np.stack([pd_new.get_covariance([np.delete(Xm, i, axis=0)])[0] for i in idx]).astype(np.float32)
    capture_cell("cell_missing_d20_k3_L15_fresh", SK, fresh, 15, 0, [14], loss_S=Sfull)
    # structure penalty (direct mode with true_theta, main.py:398,325-334)
    _, P16 = synth_X(16, 10, 920)
    capture_cell("cell_struct_d16_b1_L6_fresh", S16[:1], fresh, 6, 0, [5], struct=P16[None].astype(np.float32))

    # ---------------- consensus (main.py:673-716)
    th = rng.standard_normal((5, 12, 12)).astype(np.float32)
    th[rng.random(th.shape) < 0.3] = 0.0
    cons = uG.get_final_precision_from_batch(torch.from_numpy(th.copy()), type="min").numpy()
    np.savez_compressed(os.path.join(OUT, "consensus.npz"), theta_K=th, out_min=cons)

    # ---------------- fit: direct
    with FitRecorder() as rec, quiet():
        torch.manual_seed(7)
        g = uG.uGLAD_GL()
        g.fit(X25.copy(), centered=False, epochs=120, lr=0.002, INIT_DIAG=0, L=15, verbose=False, mode="direct")
    save_fit("fit_direct_d25", g, rec, {"X": X25, "epochs": np.int64(120), "lr": np.float64(0.002), "L": np.int64(15)})

    # ---------------- fit: cv
    X16, _ = synth_X(16, 300, 930)
    with FitRecorder() as rec, quiet():
        torch.manual_seed(8)
        g = uG.uGLAD_GL()
        g.fit(X16.copy(), epochs=30, lr=0.002, L=10, verbose=False, k_fold=3, mode="cv")
    save_fit("fit_cv_d16", g, rec, {"X": X16, "epochs": np.int64(30), "lr": np.float64(0.002), "L": np.int64(10),
                                    "k_fold": np.int64(3)})

    # ---------------- fit: missing
    X20, _ = synth_X(20, 400, 940)
    Xmiss = pd_new.add_noise_dropout(X20[None], 0.3, rng=np.random.default_rng(941))[0]
    with FitRecorder() as rec, quiet():
        torch.manual_seed(9)
        g = uG.uGLAD_GL()
        g.fit(Xmiss.copy(), epochs=50, lr=0.005, L=15, verbose=False, k_fold=3, mode="missing")
    save_fit("fit_missing_d20", g, rec, {"X": Xmiss, "epochs": np.int64(50), "lr": np.float64(0.005), "L": np.int64(15),
                                         "k_fold": np.int64(3)})

    # ---------------- fit: multitask
    Xmt = [synth_X(20, n, 950 + i)[0] for i, n in enumerate((300, 350, 400))]
    with FitRecorder() as rec, quiet():
        torch.manual_seed(10)
        g = uG.uGLAD_multitask()
        g.fit([x.copy() for x in Xmt], epochs=60, lr=0.01, L=15, verbose=False)
    extra = {"epochs": np.int64(60), "lr": np.float64(0.01), "L": np.int64(15), "n_tasks": np.int64(3)}
    for i, x in enumerate(Xmt):
        extra[f"X{i}"] = x
    save_fit("fit_multitask_d20_k3", g, rec, extra)


if __name__ == "__main__":
    main()
