#!/usr/bin/env python3
"""Adversarial inputs for the secular solver (CPU; the float32 replica of eig_lean.h's secular_root_reg in scripts/secular_study.py): tiny and
large merges, clustered poles (the gap the kernel's coupling test leaves: 4 eps max|d| + 1e-10 scale), graded poles, weights from all equal to one
dominant to the floor rho kZFloor^2, rho over six decades.  Each root against a float64 bisection of the same rational function.
Reports the worst eigenvalue error / max(|d|, rho) and the worst relative error of mu (what the Gu-Eisenstat vector needs)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import secular_study as ss
f = np.float32
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400

def exact_root(ds, rz, i):
    """root i of 1 + sum rz_j / (d_j - x) in (d_i, d_{i+1}) (or (d_last, d_last + rho sum z^2)), float64 bisection on mu = x - d_K of the nearer pole"""
    ds = ds.astype(np.float64); rz = rz.astype(np.float64); nb = len(ds)
    lo = ds[i]; hi = ds[i + 1] if i + 1 < nb else ds[i] + rz.sum() * 1.0000001 + 1e-300
    def w(x): return 1.0 + np.sum(rz / (ds - x))
    mid = 0.5 * (lo + hi)
    K = i if (i + 1 >= nb or w(mid) > 0) else i + 1  # origin: the pole nearest to the root
    a, b = (0.0, mid - ds[K]) if K == i else (mid - ds[K], 0.0)
    if i + 1 >= nb: a, b = 0.0, hi - ds[K]
    def wm(mu): return 1.0 + np.sum(rz / ((ds - ds[K]) - mu))
    for _ in range(200):
        m = 0.5 * (a + b)
        if m == a or m == b: break
        if wm(m) < 0: a = m
        else: b = m
    return K, 0.5 * (a + b)

def poles(kind, nb, scale):
    if kind == "uniform": d = np.sort(rng.uniform(-1, 1, nb))
    elif kind == "clustered":
        d = np.sort(rng.uniform(-1, 1, nb)); 
        for j in range(1, nb, 2): d[j] = d[j - 1]  # pairs pushed apart by the kernel's rule below
    elif kind == "graded": d = np.sort(10.0 ** rng.uniform(-6, 0, nb)) * rng.choice([-1, 1])
    else: d = np.sort(np.concatenate([rng.uniform(-1, -0.999, nb // 2), rng.uniform(0.999, 1, nb - nb // 2)]))
    d = np.sort(d).astype(f) * f(scale)
    return d

worst = {}
nbad = 0
for trial in range(N):
    nb = int(rng.choice([2, 3, 4, 5, 8, 16, 32]))
    kind = str(rng.choice(["uniform", "clustered", "graded", "two groups"]))
    wkind = str(rng.choice(["equal", "one dominant", "floor", "mixed"]))
    rho = f(10.0 ** rng.uniform(-3, 3))
    d = poles(kind, nb, 1.0)
    z = rng.standard_normal(nb)
    if wkind == "equal": z = np.ones(nb)
    elif wkind == "one dominant": z = 1e-4 * z; z[rng.integers(nb)] = 1.0
    elif wkind == "floor": z = np.full(nb, 1e-6) * rng.choice([-1, 1], nb); z[rng.integers(nb)] = 1.0
    z = z / np.linalg.norm(z)
    z = np.where(np.abs(z) < 1e-6, np.where(z < 0, -1e-6, 1e-6), z).astype(f)
    scale = max(np.abs(d).max(), rho)
    for j in range(1, nb):  # the kernel's "poles that need separating"
        gap = f(4 * 5.96e-8) * max(abs(d[j]), abs(d[j - 1])) + f(1e-10) * f(scale)
        if d[j] < d[j - 1] + gap: d[j] = f(d[j - 1] + gap)
    rz = (f(rho) * z * z).astype(f)
    for i in range(nb):
        ev, K, mu, _ = ss.secular_root_gragg(d, rz, rho, nb, i, relstep=2.44140625e-4)
        Kx, mux = exact_root(d, rz, i)
        lam, lamx = float(d[K]) + float(mu), float(d[Kx]) + mux
        e_abs = abs(lam - lamx) / scale
        e_mu = abs(float(mu) - mux) / max(abs(mux), 1e-300) if K == Kx else abs(lam - lamx) / max(abs(mux), 1e-300)
        key = (kind, wkind)
        w0 = worst.get(key, (0.0, 0.0, 0))
        worst[key] = (max(w0[0], e_abs), max(w0[1], e_mu), max(w0[2], ev))
        if e_abs > 5e-7 or e_mu > 1e-3:
            nbad += 1
            if nbad <= 12: print(f"trial {trial} nb {nb} {kind}/{wkind} rho {rho:.3g} root {i}: lambda error/scale {e_abs:.2e}  mu relative {e_mu:.2e}  evaluations {ev}  (K {K} vs {Kx}, mu {mu:.6g} vs {mux:.6g})")
for k, v in sorted(worst.items()): print(f"{k[0]:11s} {k[1]:13s} worst lambda error / scale {v[0]:.2e}   worst relative error of mu {v[1]:.2e}   most evaluations {v[2]}")
print("flagged roots:", nbad)
