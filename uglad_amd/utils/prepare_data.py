"""Host-side table preparation and synthetic Gaussian-graph data for the GLAD path.

This is the one-off O(M*D^2) numpy/pandas work that runs once per ``fit`` before the
unrolled GLAD cell takes over on the device.  Semantics follow the reference
(``uglad/utils/prepare_data.py``): ``process_table`` :361-516, ``normalize_table``
:597-613, ``get_covariance`` :328-356, ``convert_to_torch`` :288-307, ``get_data``
:93-140, ``add_noise_dropout`` :143-169.  Differences, all additive:

* every random draw takes an explicit ``rng`` (``numpy.random.Generator``); the reference
  consumes numpy's and Python's *global* RNG streams, which is why its notebook numbers
  are not reproducible;
* the Erdos-Renyi adjacency is drawn with numpy instead of networkx;
* nothing prints unless ``VERBOSE`` is true.
"""
from __future__ import annotations

from time import time
from typing import Optional, Sequence

import numpy as np
import torch

__all__ = [
    "generate_random_graph",
    "simulate_gaussian_samples",
    "get_data",
    "add_noise_dropout",
    "convert_to_torch",
    "eigen_val_condition_num",
    "empirical_covariance",
    "get_covariance",
    "normalize_table",
    "process_table",
    "analyse_condition_number",
    "get_highly_correlated_features",
    "synthetic_covariance_batch",
]


def _rng(rng) -> np.random.Generator:
    if isinstance(rng, np.random.Generator):
        return rng
    return np.random.default_rng(rng)


# --------------------------------------------------------------------------- synthetic
def generate_random_graph(num_nodes: int, sparsity, rng=None) -> np.ndarray:
    """Symmetric 0/1 adjacency of a G(n, p) graph, p ~ U[sparsity] (ref :13-35)."""
    rng = _rng(rng)
    lo, hi = sparsity
    p = rng.uniform(lo, hi)
    upper = np.triu(rng.random((num_nodes, num_nodes)) < p, 1)
    adj = (upper | upper.T).astype(np.float64)
    return adj


def simulate_gaussian_samples(
    num_nodes: int,
    edge_connections: np.ndarray,
    num_samples: int,
    rng=None,
    u: float = 0.1,
    w_min: float = 0.5,
    w_max: float = 1.0,
):
    """Precision = sym(adj * U[w_min,w_max]) + I, shifted so lambda_min == u; samples ~ N(0, P^-1)
    (ref :38-90).  Returns (X (num_samples, D), precision (D, D))."""
    rng = _rng(rng)
    W = rng.random((num_nodes, num_nodes)) * (w_max - w_min) + w_min
    theta = np.asarray(edge_connections, dtype=np.float64) * W
    theta = (theta + theta.T) / 2.0 + np.eye(num_nodes)
    smallest = np.linalg.eigvalsh(theta).min()
    precision = theta + np.eye(num_nodes) * (u - smallest)
    cov = np.linalg.inv(precision)
    cov = (cov + cov.T) / 2.0
    X = rng.multivariate_normal(np.zeros(num_nodes), cov, size=num_samples, method="cholesky")
    return X, precision


def get_data(
    num_nodes: int,
    sparsity,
    num_samples: int,
    batch_size: int = 1,
    w_min: float = 0.5,
    w_max: float = 1.0,
    eig_offset: float = 0.1,
    rng=None,
):
    """Batch of (samples, true precision) pairs (ref :93-140)."""
    rng = _rng(rng)
    Xb, thetas = [], []
    for _ in range(batch_size):
        adj = generate_random_graph(num_nodes, sparsity, rng)
        X, P = simulate_gaussian_samples(
            num_nodes, adj, num_samples, rng, u=eig_offset, w_min=w_min, w_max=w_max
        )
        Xb.append(X)
        thetas.append(P)
    return np.array(Xb), np.array(thetas)


def add_noise_dropout(Xb: np.ndarray, dropout: float = 0.25, rng=None) -> np.ndarray:
    """Replace a fraction of the entries of each table by NaN (ref :143-169)."""
    rng = _rng(rng)
    out = []
    for X in Xb:
        flat = np.array(X, dtype=np.float64).reshape(-1)
        idx = rng.choice(flat.size, size=int(flat.size * dropout), replace=False)
        flat[idx] = np.nan
        out.append(flat.reshape(X.shape))
    return np.array(out)


# --------------------------------------------------------------------------- conversion
def convert_to_torch(data, req_grad: bool = False, device=None) -> torch.Tensor:
    """numpy -> fp32 torch tensor (ref :288-307).  ``device`` replaces the reference's
    ``use_cuda`` flag; default stays host memory like the reference."""
    if not torch.is_tensor(data):
        data = torch.from_numpy(np.asarray(data).astype(np.float64, copy=False)).to(torch.float32)
    if device is not None:
        data = data.to(device)
    data.requires_grad = req_grad
    return data


# --------------------------------------------------------------------------- covariance
def empirical_covariance(X, assume_centered: bool = False) -> np.ndarray:
    """Maximum-likelihood covariance X^T X / n, as sklearn.covariance.empirical_covariance
    (the routine the reference calls at :342 and main.py:139)."""
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X.reshape(1, -1)
    if assume_centered:
        return X.T @ X / X.shape[0]
    return np.cov(X.T, bias=1).reshape(X.shape[1], X.shape[1])


def eigen_val_condition_num(A: np.ndarray):
    """Real parts of the eigenvalues and max|eig|/min|eig| (ref :310-325)."""
    eig = [float(v.real) for v in np.linalg.eigvals(A)]
    a = np.abs(eig)
    return eig, float(a.max() / a.min())


def get_covariance(Xb, offset: float = 0.1, VERBOSE: bool = False) -> np.ndarray:
    """Batch covariance with the reference's eigenvalue repair (ref :328-356): when the
    smallest eigenvalue is <= 1e-6 the matrix is shifted so that it becomes ``offset``."""
    Sb = []
    for X in Xb:
        S = empirical_covariance(X, assume_centered=False)
        eig, con = eigen_val_condition_num(S)
        if min(eig) <= 1e-6:
            if VERBOSE:
                print(f"Adjust the eval: min {min(eig)}, con {con}")
            S = S + np.eye(S.shape[-1]) * (offset - min(eig))
        Sb.append(S)
    return np.array(Sb)


# --------------------------------------------------------------------------- table checks
def normalize_table(df, typeN: str):
    """'min_max' | 'mean' | anything else = untouched (ref :597-613)."""
    if typeN == "min_max":
        return (df - df.min()) / (df.max() - df.min())
    if typeN == "mean":
        return (df - df.mean()) / df.std()
    return df


def analyse_condition_number(table, MESSAGE: str = "", VERBOSE: bool = True):
    """Covariance, eigenvalues and condition number of a table (ref :569-594)."""
    S = empirical_covariance(np.asarray(table, dtype=np.float64), assume_centered=False)
    eig, con = eigen_val_condition_num(S)
    if VERBOSE:
        print(f"{MESSAGE} covariance matrix: condition number {con}, min eig {min(eig)} max eig {max(eig)}")
    return S, eig, con


def get_highly_correlated_features(input_cov: np.ndarray) -> np.ndarray:
    """Rank features by how many top-10% |cov-of-cov| partners they have (ref :519-550)."""
    cov2 = empirical_covariance(input_cov)
    np.fill_diagonal(cov2, 0.0)
    a = np.abs(cov2)
    r, c = np.triu_indices(a.shape[0], 1)
    upper = np.sort(a[r, c])[::-1]
    th = upper[int(0.1 * len(upper))]
    rows, _ = np.nonzero(a >= th)
    feats, counts = np.unique(rows, return_counts=True)
    order = np.argsort(-counts, kind="stable")
    return feats[order]


def process_table(
    table,
    NORM: str = "no",
    MIN_VARIANCE: float = 0.0,
    msg: str = "",
    COND_NUM: float = np.inf,
    eigval_th: float = 1e-3,
    VERBOSE: bool = True,
):
    """Make a real-valued table fit for sparse graph recovery (ref :361-516), in the
    reference's order: drop all-zero rows, fill NaN with the column mean, drop
    single-valued columns, normalise, drop duplicate columns, drop low-variance columns,
    then (only when ``COND_NUM`` is finite) drop highly correlated columns until the
    covariance condition number is acceptable.  Returns a pandas DataFrame."""
    import pandas as pd

    start = time()
    table = pd.DataFrame(table).astype(float)
    n0 = table.shape[0]
    table = table.loc[~(table == 0).all(axis=1)]
    if VERBOSE:
        print(f"{msg}: input {n0} samples x {table.shape[1]} features; zero rows dropped {n0 - table.shape[0]}")
    table = table.fillna(table.mean())
    single = [c for c in table.columns if table[c].nunique(dropna=False) == 1]
    table = table.drop(columns=single)
    table = normalize_table(table, NORM)
    if VERBOSE:
        print(f"{msg}: single-valued columns dropped {len(single)}")
        analyse_condition_number(table, "Input", VERBOSE)
    cols = table.columns
    table = table.T.drop_duplicates().T
    if VERBOSE:
        print(f"{msg}: duplicate columns dropped {len(cols) - len(table.columns)}")
    var = table.var()
    low = list(var[var < MIN_VARIANCE].index)
    table = table.drop(columns=low)
    cov_table, eig, con = analyse_condition_number(table, "Processed", VERBOSE)
    itr = 1
    while con > COND_NUM:
        lb = int(np.sum(np.array(eig) < eigval_th))
        if lb == 0:
            lb = 1
        feats = get_highly_correlated_features(cov_table)
        feats = feats[: min(lb, len(feats))]
        table = table.drop(columns=table.columns[feats])
        cov_table, eig, con = analyse_condition_number(table, f"{msg} {itr}: corr dropped", VERBOSE)
        itr += 1
    if VERBOSE:
        print(f"{msg}: processed table {table.shape[0]} x {table.shape[1]} in {np.round(time() - start, 3)} s")
    return table


# --------------------------------------------------------------------------- bench inputs
def synthetic_covariance_batch(
    num_tasks: int,
    num_nodes: int,
    num_samples: Optional[int] = None,
    seed: int = 1234,
    sparsity: Sequence[float] = (0.1, 0.2),
    eig_offset: float = 1.0,
    task_offset: int = 0,
) -> np.ndarray:
    """The benchmark/parity input of SURVEY.md section 8(d): per task draw a Gaussian graph,
    sample it, min-max normalise the columns (what ``fit`` always does, ref main.py:85) and take
    the repaired empirical covariance.  Task ``i`` uses ``default_rng(seed + task_offset + i)``
    so a sharded rank generates exactly its slice of the global batch.  Returns fp32 (K, D, D)."""
    if num_samples is None:
        num_samples = 1024 if num_nodes >= 256 else 500
    out = np.empty((num_tasks, num_nodes, num_nodes), dtype=np.float32)
    for i in range(num_tasks):
        rng = np.random.default_rng(seed + task_offset + i)
        Xb, _ = get_data(num_nodes, sparsity, num_samples, 1, eig_offset=eig_offset, rng=rng)
        X = Xb[0]
        X = (X - X.min(0)) / (X.max(0) - X.min(0))
        out[i] = get_covariance([X], offset=0.1)[0].astype(np.float32)
    return out
