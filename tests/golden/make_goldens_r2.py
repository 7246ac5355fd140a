#!/usr/bin/env python3
"""Round-2 additions to the golden vectors: more outputs of the REAL reference, captured in the build container.

Same recipe as make_goldens.py; run from a scratch directory:

    cd /tmp && python /root/repo/tests/golden/make_goldens_r2.py

Adds (inputs AND expected outputs, never reference source):
  cell_d129_b2_L30_trained.npz   odd D just past the LDS-resident size (NT = 5 path)
  cell_d200_b1_L30_trained.npz   D between the 32-multiples of the workspace-resident path (NT = 7)
  cell_d256_b1_L30_trained.npz   the largest supported D with trained parameters
  fit_direct_d25_converged.npz   uGLAD_GL.fit(mode="direct") run to convergence (600 epochs): loss per epoch, precision_
  map_d25.npz, map_d64.npz       conditional_gaussian_with_probabilities / compute_map_estimate (main.py:1176-1260)
  metrics_k3_d20.npz             report_metrics_all (utils/metrics.py:25-108) + get_partial_correlations (main.py:796-821)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg  # noqa: E402  (sets up the reference import: pyvis stub, Agg backend, sys.path)

import numpy as np  # noqa: E402
import torch  # noqa: E402

uG = mg.uG
OUT = mg.OUT
METRIC_KEYS = ("FDR", "TPR", "FPR", "SHD", "nnzTrue", "nnzPred", "precision", "recall", "Fbeta", "aupr", "auc")


def capture_map(name, precision, mean, observed_idx, observed_values, node_names=None):
    full_mean, cond_cov, pdf = uG.conditional_gaussian_with_probabilities(
        np.asarray(precision), np.asarray(mean), list(observed_idx), np.asarray(observed_values, dtype=np.float64))
    out = {
        "precision": np.asarray(precision),
        "mean": np.asarray(mean),
        "observed_idx": np.asarray(observed_idx, dtype=np.int64),
        "observed_values": np.asarray(observed_values, dtype=np.float64),
        "full_mean": np.asarray(full_mean, dtype=np.float64),
        "cond_cov": np.asarray(cond_cov, dtype=np.float64),
        "pdf": np.float64(pdf),
    }
    if node_names is not None:
        class _M:  # what compute_map_estimate reads from a fitted estimator
            pass

        m = _M()
        m.precision_, m.location_, m.node_names_ = np.asarray(precision), np.asarray(mean), list(node_names)
        with mg.quiet():
            out["map_clipped"] = np.asarray(uG.compute_map_estimate(
                {node_names[i]: float(v) for i, v in zip(observed_idx, observed_values)}, m), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: D={len(mean)} observed {list(observed_idx)} pdf {pdf:.4e}")


def main():
    os.chdir("/tmp")
    trained = {k: np.array(v) for k, v in np.load(os.path.join(OUT, "params_trained.npz")).items()}

    # ---------------- cells beyond the LDS-resident size
    mg.capture_cell("cell_d129_b2_L30_trained", mg.synth_S(2, 129, 1290), trained, 30, 0, [29], keep_init=False)
    mg.capture_cell("cell_d200_b1_L30_trained", mg.synth_S(1, 200, 2000), trained, 30, 0, [29], keep_init=False)
    mg.capture_cell("cell_d256_b1_L30_trained", mg.synth_S(1, 256, 901), trained, 30, 0, [], keep_init=False)

    # ---------------- fit: direct, to convergence (SURVEY.md section 7 hard part 2: assert precision_ at convergence)
    X25, P25 = mg.synth_X(25, 500, 77)
    with mg.FitRecorder() as rec, mg.quiet():
        torch.manual_seed(7)
        g = uG.uGLAD_GL()
        g.fit(X25.copy(), centered=False, epochs=600, lr=0.002, INIT_DIAG=0, L=15, verbose=False, mode="direct")
    mg.save_fit("fit_direct_d25_converged", g, rec,
                {"X": X25, "epochs": np.int64(600), "lr": np.float64(0.002), "L": np.int64(15), "true_theta": P25})

    # ---------------- MAP / conditional Gaussian (main.py:1176-1260)
    names = [f"node_{i}" for i in range(25)]
    capture_map("map_d25", g.precision_, g.location_, [3, 10, 17], [0.7, 0.2, 0.55], node_names=names)
    rng = np.random.default_rng(6400)
    A = rng.standard_normal((64, 64))
    P64 = A @ A.T / 64 + 0.5 * np.eye(64)
    P64[np.abs(P64) < 0.08] = 0.0
    P64 = 0.5 * (P64 + P64.T) + 0.3 * np.eye(64)
    obs = sorted(rng.choice(64, size=20, replace=False).tolist())
    capture_map("map_d64", P64, rng.random(64), obs, rng.random(20), node_names=[f"n{i}" for i in range(64)])

    # ---------------- metrics + partial correlations
    true_K, pred_K, met_K, pc_K = [], [], [], []
    for i in range(3):
        _, P = mg.synth_X(20, 10, 3000 + i)
        r = np.random.default_rng(3100 + i)
        pred = P + (0.25 + 0.1 * i) * r.standard_normal(P.shape)
        pred = 0.5 * (pred + pred.T)
        pred[np.abs(pred) < 0.3] = 0.0  # exact zeros, as the soft threshold leaves them
        np.fill_diagonal(pred, np.abs(np.diag(P)) + 0.5)
        m = uG.report_metrics_all(P, pred)
        true_K.append(P)
        pred_K.append(pred)
        met_K.append([m[k] for k in METRIC_KEYS])
        pc_K.append(uG.get_partial_correlations(pred))
    np.savez_compressed(os.path.join(OUT, "metrics_k3_d20.npz"), true_theta=np.array(true_K), pred_theta=np.array(pred_K),
                        metrics=np.array(met_K, dtype=np.float64), partial_correlations=np.array(pc_K))
    print("metrics_k3_d20:", dict(zip(METRIC_KEYS, met_K[0])))


if __name__ == "__main__":
    main()
