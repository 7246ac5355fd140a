#!/usr/bin/env python3
"""Diagnostic (GPU box): does the tridiagonalisation of chunk c+1 overlap with the D&C/cell kernel of chunk c when the two
run on different streams?  Times L forward cell steps sequentially and chunk-pipelined."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uglad_amd
from uglad_amd import _lib
from uglad_amd.utils.prepare_data import synthetic_covariance_batch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
L = 30
lib = _lib.get_lib()
base = synthetic_covariance_batch(16, D, seed=5)
S = torch.from_numpy(np.tile(base, (M // 16 + 1, 1, 1))[:M]).cuda().contiguous()
model = uglad_amd.GladParams(1.0, device="cuda:0")
params = model.packed().detach()
DP = ((D + 31) // 32) * 32
Z = [torch.empty_like(S), torch.empty_like(S)]
half = torch.empty_like(S); U = torch.empty_like(S); beta = torch.empty(M, D, device="cuda")
nf = torch.empty(M, device="cuda"); nsum = torch.empty(1, device="cuda")
lam = torch.full((L + 1,), 0.5, device="cuda"); lam_in = torch.empty(L + 1, 2, device="cuda")
wsp = lib.workspace(M, D, S)
lib.init_theta(S, params, 0, Z[0], wsp)
torch.cuda.synchronize()
theta0 = Z[0].clone()


def run_seq():
    Z[0].copy_(theta0)
    for k in range(L):
        lib.cell_fwd(S, Z[k % 2], lam[k:], params, Z[(k + 1) % 2], half, U, beta, nf, wsp, 1)
        lib.sum_partials(nf, nsum)


aux = torch.cuda.Stream()


def run_pipe(chunk, two_streams=True):
    main = torch.cuda.current_stream()
    Z[0].copy_(theta0)
    nch = (M + chunk - 1) // chunk
    for k in range(L):
        zi, zo = Z[k % 2], Z[(k + 1) % 2]
        aux.wait_stream(main)
        for c in range(nch):
            a, b = c * chunk, min(M, (c + 1) * chunk)
            w = wsp[a * 3 * DP:]
            if two_streams:
                with torch.cuda.stream(aux):
                    lib.tridiagonalize(S[a:b], zi[a:b], lam[k:], zo[a:b], w)
                    ev = torch.cuda.Event()
                    ev.record(aux)
                main.wait_event(ev)
            else:
                lib.tridiagonalize(S[a:b], zi[a:b], lam[k:], zo[a:b], w)
            lib.cell_fwd_stage2(S[a:b], zi[a:b], lam[k:], params, zo[a:b], half[a:b], U[a:b], beta[a:b], nf[a:], w, 1)
        lib.sum_partials(nf, nsum)


def timeit(fn, *a):
    fn(*a); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn(*a)
    th = (time.perf_counter() - t0) / 3 * 1e3
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3 * 1e3, th


t_seq, _ = timeit(run_seq)
ref = Z[L % 2].clone()
print(f"M={M} D={D} L={L}: sequential {t_seq:.2f} ms/pass ({t_seq / L:.3f} ms/step)", flush=True)
for chunk in (512, 256, 128, 64):
    if chunk >= M:
        continue
    t, th = timeit(run_pipe, chunk)
    err = float((Z[L % 2] - ref).abs().max())
    t1, th1 = timeit(run_pipe, chunk, False)
    print(f"  chunk={chunk:4d}: two streams {t:.2f} ms/pass (host {th:.2f}), one stream {t1:.2f} (host {th1:.2f})  max|dTheta| vs sequential {err:.1e}", flush=True)
