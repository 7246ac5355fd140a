#!/usr/bin/env python3
"""The reference's OWN fp32 noise floor on the 42 parameter gradients, per golden and parameter tensor.

For every cell golden (outputs of the real reference, fp32) the same function -- the reference's 10-step Newton-Schulz
forward and backward applied per eigenvalue, oracle/glad_exact.py mode "ns10" -- is evaluated in fp64; the relative
distance of the reference's gradient from that fp64 value is what fp32 arithmetic costs the reference itself.  A kernel
cannot be asked to sit closer to the reference than the reference sits to its own exact-arithmetic value.

    python tests/golden/measure_grad_noise.py [golden names ...]     # CPU, the build container; writes grad_noise_floor.json

tests/test_gpu_parity.py bounds the kernels' gradient error by max(1e-4, 2 x this floor): a fixed table, nothing in it comes from the
build under test (round 2 merged observed errors into a tolerance table; round 3 removed that).
"""
import glob
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import glad_exact as ex  # noqa: E402


def relF(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def noise_floor(only=None):
    out = {}
    for path in sorted(glob.glob(os.path.join(HERE, "cell_*.npz")) + glob.glob(os.path.join(HERE, "regime_*.npz"))):
        name = os.path.basename(path)[:-4]
        if only and name not in only:
            continue
        g = np.load(path)
        p = ex.params64(g, "param.")
        L, diag = int(g["L"]), int(g["INIT_DIAG"])
        loss_S = g["loss_S"] if "loss_S" in g else None
        struct = g["struct"] if "struct" in g else None
        theta, tr = ex.glad_forward(g["S"], p, L, diag, loss_S=loss_S, struct=struct, mode="ns10")
        grads = ex.glad_backward(g["S"], p, L, tr, diag, loss_S=loss_S, struct=struct, mode="ns10")
        out[name] = {"theta_relF": max(relF(theta[i], g["theta_L"][i]) for i in range(theta.shape[0])),
                     "grads": {k: relF(g["grad." + k], grads[k]) for k in ex.PARAM_KEYS}}
        worst = max(out[name]["grads"].items(), key=lambda kv: kv[1])
        print(f"{name:34s} Theta {out[name]['theta_relF']:.2e}   worst gradient {worst[0]} {worst[1]:.2e}", flush=True)
    return out


def main():
    """No arguments: every golden.  With golden names: only those, merged into the existing table."""
    floor_path = os.path.join(HERE, "grad_noise_floor.json")
    only = set(sys.argv[1:])
    table = json.load(open(floor_path)) if only and os.path.exists(floor_path) else {}
    table.update(noise_floor(only))
    json.dump(table, open(floor_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
