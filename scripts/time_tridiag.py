#!/usr/bin/env python3
"""Diagnostic (GPU box): launch time of tridiag_kernel alone and its chain / sweep split (stamps build: UGLAD_DIAG_SO or scripts/_build/libuglad_diag.so)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.environ.get("UGLAD_DIAG_SO", os.path.join(ROOT, "scripts", "_build", "libuglad_diag.so"))
from uglad_amd import _lib
from uglad_amd.utils.prepare_data import synthetic_covariance_batch
D = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lib = _lib.HipLib(so, require_gpu=True)
base = synthetic_covariance_batch(8, D, seed=5)
S = torch.from_numpy(np.tile(base, (M // 8 + 1, 1, 1))[:M]).cuda().contiguous()
Z = torch.eye(D, device="cuda").repeat(M, 1, 1).contiguous(); R = torch.empty_like(S)
lam = torch.full((1,), 0.5, device="cuda"); wsp = lib.workspace(M, D, S)
for _ in range(3):
    lib.tridiagonalize(S, Z, lam, R, wsp)
torch.cuda.synchronize()
tb = (ctypes.c_ulonglong * 4)()
try:
    lib._dll.uglad_diag_tstamps(ctypes.cast(tb, ctypes.c_void_p), 1)
    lib.tridiagonalize(S, Z, lam, R, wsp); torch.cuda.synchronize()
    lib._dll.uglad_diag_tstamps(ctypes.cast(tb, ctypes.c_void_p), 1)
    print(f"workgroup 0, cycles over all steps: reflector chain + barrier {tb[0]}  sweep + barrier {tb[1]}  prologue {tb[2]}")
except AttributeError:
    pass
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    lib.tridiagonalize(S, Z, lam, R, wsp)
e1.record(); torch.cuda.synchronize()
print(f"tridiag D={D} M={M}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch   [{so}]")
