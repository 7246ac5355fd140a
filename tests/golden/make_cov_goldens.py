#!/usr/bin/env python3
"""Golden vectors for the covariance front-end, made by the REAL reference on CPU (build container only):
tables X -> normalize_table(min_max) -> get_covariance(offset) of /root/reference/uglad/utils/prepare_data.py.

    cd /tmp && python /root/repo/tests/golden/make_cov_goldens.py

Stored per case: X (K,N,D) fp32 raw tables, Xn (normalised, fp64; small cases only), S (K,D,D) fp64 exactly as the reference returns it
(repaired where its smallest eigenvalue was <= 1e-6), S_raw (before the repair; small cases only), offset.  Data only, no reference source."""
import contextlib
import io
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
from sklearn import covariance  # noqa: E402

from uglad.utils import prepare_data as ref  # noqa: E402  (the reference)

OUT = os.path.dirname(os.path.abspath(__file__))


def make(name, K, N, D, seed, offset=0.1, rank=None):
    rng = np.random.default_rng(seed)
    Xs = []
    for _ in range(K):
        A = rng.standard_normal((D, D)) / np.sqrt(D) + np.eye(D)
        X = rng.standard_normal((N, D)) @ A
        if rank is not None:  # columns beyond `rank` are combinations of the first ones: a singular covariance
            X[:, rank:] = X[:, :rank] @ rng.standard_normal((rank, D - rank))
        X = X * rng.uniform(0.5, 20.0, size=D) + rng.uniform(-5, 5, size=D)  # columns on very different scales
        Xs.append(X.astype(np.float32))
    X = np.stack(Xs)
    Xn = np.stack([np.array(ref.normalize_table(pd.DataFrame(x.astype(np.float64)), "min_max")) for x in X])
    S_raw = np.stack([covariance.empirical_covariance(x, assume_centered=False) for x in Xn])
    with contextlib.redirect_stdout(io.StringIO()):
        S = ref.get_covariance(Xn, offset=offset)
    extra = {"Xn": Xn, "S_raw": S_raw} if Xn.size <= 70_000 else {}  # (only for the small cases: fixture size)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), X=X, S=S, offset=np.float64(offset), **extra)
    rep = [bool(np.abs(S[k] - S_raw[k]).max() > 0) for k in range(K)]
    print(name, X.shape, "repaired:", rep)


make("cov_k3_n500_d20", 3, 500, 20, 11)
make("cov_k2_n40_d64_singular", 2, 40, 64, 12)            # N < D: rank-deficient -> eigenvalue repair
make("cov_k2_n300_d128_rank100", 2, 300, 128, 13, rank=100)
make("cov_k1_n512_d256", 1, 512, 256, 14)
make("cov_k2_n97_d33", 2, 97, 33, 15, offset=0.25)        # ragged sizes (N not a multiple of the chunk, D not of 32)
