#!/usr/bin/env python3
"""Randomised sweep of the matrix-iteration path on the GPU: sizes, batch sizes, sample counts (conditioning), parameter sets; the result
against the fp64 oracle.  Not part of the suite (a minute of oracle time); prints one line per case and the worst.
    python tests/experiments/ns_fuzz.py [cases]"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
import uglad_amd  # noqa: E402
from oracle import glad_exact as ex  # noqa: E402
from uglad_amd.utils.prepare_data import synthetic_covariance_batch  # noqa: E402

GOLDEN = os.path.join(HERE, "..", "golden")


def relF(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(7)
    worst = (0.0, None)
    for case in range(n):
        D = int(rng.choice([rng.integers(129, 257), rng.integers(257, 513), rng.integers(513, 801)]))
        B = int(rng.integers(1, 4))
        N = int(D * rng.choice([1.5, 2, 4]))
        which = rng.choice(["fresh", "trained"])
        L = int(rng.integers(1, 4))
        g = np.load(os.path.join(GOLDEN, f"params_{which}.npz"))
        m = uglad_amd.GladParams(1.0, device="cuda")
        m.load_state_dict({k: torch.from_numpy(np.array(g[k])) for k in ex.PARAM_KEYS})
        S = synthetic_covariance_batch(B, D, N, seed=1000 + case)
        theta, loss = uglad_amd.forward_uGLAD(torch.from_numpy(S).cuda(), m, L=L)
        loss.backward()
        p = ex.params64(g, "")
        ref, tr = ex.glad_forward(S, p, L, 0, mode="ns10")
        grads = ex.glad_backward(S, p, L, tr, 0, mode="ns10")
        sd = dict(m.named_parameters())
        et = max(relF(theta[i].detach().cpu().numpy(), ref[i]) for i in range(B))
        eg = max(relF(sd[k].grad.cpu().numpy(), grads[k]) for k in ex.PARAM_KEYS)
        el = abs(loss.item() - tr["loss"]) / max(1.0, abs(tr["loss"]))
        sym = bool(torch.equal(theta, theta.transpose(1, 2)))
        print(f"case {case:2d}: D={D:4d} B={B} N={N:5d} {which:7s} L={L}: Theta {et:.1e}  loss {el:.1e}  worst grad {eg:.1e}  symmetric {sym}", flush=True)
        if max(et, eg) > worst[0] or not np.isfinite(et + eg):
            worst = (max(et, eg), case)
    print("worst:", worst)


if __name__ == "__main__":
    main()
