#!/usr/bin/env python3
"""Round-2 additions (second batch) to the golden vectors: outputs of the REAL reference for the padded sizes that had no cell-level
case yet.  Same recipe as make_goldens.py; run from a scratch directory:

    cd /tmp && python /root/repo/tests/golden/make_goldens_r2b.py

Adds (inputs AND expected outputs, never reference source):
  cell_d96_b2_L30_trained.npz   NT = 3 (65..96): the one instantiation without a cell golden
  cell_d33_b3_L15_fresh.npz     first size of NT = 2, odd D (no 16-byte vector paths), fresh parameters
  cell_d3_b2_L6_trained.npz     tiny: the solver's trailing 2 x 2 block right after the first reflector
  cell_d2_b2_L6_fresh.npz       the smallest matrix with an off-diagonal entry (no reflector at all)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg  # noqa: E402  (sets up the reference import: pyvis stub, Agg backend, sys.path)

import numpy as np  # noqa: E402

OUT = mg.OUT


def main():
    os.chdir("/tmp")
    trained = {k: np.array(v) for k, v in np.load(os.path.join(OUT, "params_trained.npz")).items()}
    fresh = {k: np.array(v) for k, v in np.load(os.path.join(OUT, "params_fresh.npz")).items()}
    mg.capture_cell("cell_d96_b2_L30_trained", mg.synth_S(2, 96, 960), trained, 30, 0, [29], keep_init=False)
    mg.capture_cell("cell_d33_b3_L15_fresh", mg.synth_S(3, 33, 330), fresh, 15, 0, [14], keep_init=False)
    mg.capture_cell("cell_d3_b2_L6_trained", mg.synth_S(2, 3, 30), trained, 6, 0, [5], keep_init=True)
    mg.capture_cell("cell_d2_b2_L6_fresh", mg.synth_S(2, 2, 20), fresh, 6, 0, [5], keep_init=True)


if __name__ == "__main__":
    main()
