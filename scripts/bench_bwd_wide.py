#!/usr/bin/env python3
"""Diagnostic (GPU box): one backward-cell call at D > 128 for several batch sizes; run once with UGLAD_WIDE_BWD=0 and once with =1."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uglad_amd import _lib
from uglad_amd.utils.prepare_data import synthetic_covariance_batch
lib = _lib.get_lib()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pz = np.load(os.path.join(ROOT, "tests", "golden", "params_trained.npz"))
pk = torch.tensor(np.concatenate([pz[k].ravel() for k in pz.files]), dtype=torch.float32, device="cuda")
f32 = dict(dtype=torch.float32, device="cuda")
base = synthetic_covariance_batch(8, D, seed=5)
for M in [int(x) for x in sys.argv[2:]] or [1, 8, 32, 64, 128, 256, 512]:
    S = torch.from_numpy(np.tile(base, (M // 8 + 1, 1, 1))[:M]).cuda().contiguous()
    Z0, Z1, half, U, G, Go = (torch.empty(M, D, D, **f32) for _ in range(6))
    beta, nfp = torch.empty(M, D, **f32), torch.empty(M, **f32)
    lam, lam_in = torch.empty(2, **f32), torch.empty(2, 2, **f32)
    wsp = lib.workspace(M, D, S)
    lib.init_theta(S, pk, 0, Z0, wsp); lib.lambda_init(pk, 1.0, lam[0:1], lam_in[0])
    lib.cell_fwd(S, Z0, lam[0:1], pk, Z1, half, U, beta, nfp, wsp, 1)
    G.copy_(torch.randn(M, D, D, **f32)); G.copy_(G + G.transpose(1, 2))
    grp, glp = torch.zeros(M, 28, **f32), torch.zeros(M, **f32)
    for _ in range(2):
        lib.cell_bwd(G, S, Z0, half, U, beta, lam[0:1], pk, Go, grp, glp, 1, wsp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.cell_bwd(G, S, Z0, half, U, beta, lam[0:1], pk, Go, grp, glp, 1, wsp)
    e1.record(); torch.cuda.synchronize()
    print(f"D={D} M={M:5d} wide={os.environ.get('UGLAD_WIDE_BWD', 'auto')}: cell_bwd {e0.elapsed_time(e1) / 10 * 1e3:9.1f} us", flush=True)
