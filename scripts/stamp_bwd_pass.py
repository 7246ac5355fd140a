#!/usr/bin/env python3
"""Diagnostic (GPU box): the whole backward pass (L steps in ONE launch for D <= 128) on a -DUGLAD_STAMPS build: time per pass and the
phase ticks of workgroup 0 in its LAST step (not the first, so without the G_next load; with the copy-out of dL/dZ_0).
UGLAD_PERSISTENT_BWD=0: one launch per step.    python scripts/stamp_bwd_pass.py [D] [M] [L]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.environ.get("UGLAD_DIAG_SO", os.path.join(ROOT, "scripts", "_build", "libuglad_diag.so"))
from uglad_amd import _lib
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
L = int(sys.argv[3]) if len(sys.argv) > 3 else 30
_lib._SIGS["uglad_diag_kstamps"] = ([ctypes.c_void_p], ctypes.c_int)
lib = _lib.HipLib(so, require_gpu=True)
from uglad_amd.utils.prepare_data import synthetic_covariance_batch
base = synthetic_covariance_batch(8, D, seed=5)
S = torch.from_numpy(np.tile(base, (M // 8 + 1, 1, 1))[:M]).cuda().contiguous()
pz = np.load(os.path.join(ROOT, "tests", "golden", "params_trained.npz"))
pk = torch.tensor(np.concatenate([pz[k].ravel() for k in pz.files]), dtype=torch.float32, device="cuda")
f32 = dict(dtype=torch.float32, device="cuda")
Z = torch.empty(L + 1, M, D, D, **f32)
half, U = torch.empty(L, M, D, D, **f32), torch.empty(L, M, D, D, **f32)
beta = torch.empty(L, M, D, **f32)
lam, lam_in = torch.empty(L + 1, **f32), torch.empty(L + 1, 2, **f32)
nfp, nfs = torch.empty(M, **f32), torch.empty(1, **f32)
wsp = lib.workspace(M, D, S)
lib.glad_forward(S, pk, 1.0, 0, L, Z, half, U, beta, lam, lam_in, nfp, nfs, wsp, 1)
G = torch.randn(M, D, D, generator=torch.Generator(device="cuda").manual_seed(1), **f32)
G = (G + G.transpose(1, 2)).contiguous()
g0, g1 = torch.empty_like(S), torch.empty_like(S)
grp, glp, gtp, grad = torch.empty(M, 28, **f32), torch.empty(L, M, **f32), torch.empty(M, **f32), torch.empty(42, **f32)
def run():
    lib.glad_backward(G, S, pk, 0, L, Z, half, U, beta, lam, lam_in, g0, g1, grp, glp, gtp, grad, wsp, 1)
for _ in range(2):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"backward pass D={D} M={M} L={L} persistent={os.environ.get('UGLAD_PERSISTENT_BWD', '1')}: {ms:.3f} ms per pass = {ms / L * 1e3:.1f} us per step (incl. init_bwd, finish)")
kb = (ctypes.c_ulonglong * 32)()
assert lib._dll.uglad_diag_kstamps(ctypes.cast(kb, ctypes.c_void_p)) == 0
k = np.array(list(kb), dtype=np.int64)
names = ["load U + spectrum", "phase A (rhoNN bwd)", "gemm 1", "gemm 2", "C o F", "gemm 3", "gemm 4", "G_out", "reductions"]
print("workgroup 0, last step (s_memtime ticks): " + "  ".join(f"{n} {int(k[i+1]-k[i])}" for i, n in enumerate(names)) + f"  total {int(k[9]-k[0])}")
print(f"inside G_out: store product {int(k[20]-k[7])}  assemble {int(k[21]-k[20])}  copy-out {int(k[8]-k[21])}")
print(f"grad checksum {float(grad.double().sum()):.6e}")
