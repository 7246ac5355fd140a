#!/usr/bin/env python3
"""Round 3, D > 256: goldens from the REAL reference for the sizes the matrix-iteration path serves (uglad_amd/csrc/wide_ns.h).

    cd /tmp && python /root/repo/tests/golden/make_goldens_r3_large.py

Same capture as make_goldens.py::capture_cell (inputs, lambda sequence, Theta_L, loss, the 42 gradients; no intermediates to keep the
files small).  Inputs come from this repo's seeded generator, every output from the reference's own glad()/forward_uGLAD().
  cell_d320_b1_L15_trained.npz   D = 320: not a multiple of 64 (ragged 64 x 64 tiles), 5 x 64
  cell_d512_b1_L15_trained.npz   D = 512 (the reference's Theta_L is indefinite there)
  cell_d288_b2_L6_fresh.npz      two matrices, freshly initialised parameters
  fit_direct_d288.npz            uGLAD_GL.fit(mode="direct") on a 400 x 288 table, 10 epochs (fewer divide by zero in the reference: main.py:409), L = 10
  cell_d1024_b1_L6_fresh.npz     D = 1024 (`... d1024`)
`python make_goldens_r3_large.py fit` makes the fit golden only.
"""
import os
import sys

import numpy as np
import torch

import make_goldens as mg


def main():
    os.chdir("/tmp")
    trained = {k: np.array(v) for k, v in np.load(os.path.join(mg.OUT, "params_trained.npz")).items()}
    fresh = {k: np.array(v) for k, v in np.load(os.path.join(mg.OUT, "params_fresh.npz")).items()}
    # one thread: this container's MKL hangs in the multi-threaded batched LU (torch.inverse of a (2, 288, 288) tensor: "Parameter 6 was
    # incorrect on entry to SLASWP", then no progress); single-threaded the same call returns in milliseconds, error 2.5e-6
    torch.set_num_threads(1)
    if "d1024" in sys.argv[1:]:  # the advertised maximum since the factorisation takes 1024: fresh parameters, six steps
        mg.capture_cell("cell_d1024_b1_L6_fresh", mg.synth_S(1, 1024, 10240), fresh, 6, 0, [], keep_init=False)
        return
    if "fit" not in sys.argv[1:]:
        cells(trained, fresh)
    X, _ = mg.synth_X(288, 400, 2881)
    with mg.FitRecorder() as rec, mg.quiet():
        torch.manual_seed(11)
        g = mg.uG.uGLAD_GL()
        g.fit(X.copy(), centered=False, epochs=10, lr=0.002, INIT_DIAG=0, L=10, verbose=False, mode="direct")
    mg.save_fit("fit_direct_d288", g, rec, {"X": X, "epochs": np.int64(10), "lr": np.float64(0.002), "L": np.int64(10)})


def cells(trained, fresh):
    mg.capture_cell("cell_d288_b2_L6_fresh", mg.synth_S(2, 288, 2880), fresh, 6, 0, [], keep_init=False)
    mg.capture_cell("cell_d320_b1_L15_trained", mg.synth_S(1, 320, 3200), trained, 15, 0, [], keep_init=False)
    mg.capture_cell("cell_d512_b1_L15_trained", mg.synth_S(1, 512, 5120), trained, 15, 0, [], keep_init=False)


if __name__ == "__main__":
    main()
