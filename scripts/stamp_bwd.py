#!/usr/bin/env python3
"""Diagnostic (GPU box): time of one backward-cell launch and the phase cycles of its workgroup 0, from a -DUGLAD_STAMPS build made
by scripts/dev_build.sh (scripts/_build/libuglad_diag.so).  UGLAD_LEAN_BWD=0 selects the round-1 kernel (A/B)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.environ.get("UGLAD_DIAG_SO", os.path.join(ROOT, "scripts", "_build", "libuglad_diag.so"))
from uglad_amd import _lib
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
_lib._SIGS["uglad_diag_kstamps"] = ([ctypes.c_void_p], ctypes.c_int)
lib = _lib.HipLib(so, require_gpu=True)
from uglad_amd.utils.prepare_data import synthetic_covariance_batch
base = synthetic_covariance_batch(8, D, seed=5)
S = torch.from_numpy(np.tile(base, (M // 8 + 1, 1, 1))[:M]).cuda().contiguous()
pz = np.load(os.path.join(ROOT, "tests", "golden", "params_trained.npz"))
pk = torch.tensor(np.concatenate([pz[k].ravel() for k in pz.files]), dtype=torch.float32, device="cuda")
f32 = dict(dtype=torch.float32, device="cuda")
Z0, Z1, half, U, G, Go = (torch.empty(M, D, D, **f32) for _ in range(6))
beta, nfp = torch.empty(M, D, **f32), torch.empty(M, **f32)
lam, lam_in = torch.empty(2, **f32), torch.empty(2, 2, **f32)
wsp = lib.workspace(M, D, S)
lib.init_theta(S, pk, 0, Z0, wsp); lib.lambda_init(pk, 1.0, lam[0:1], lam_in[0])
lib.cell_fwd(S, Z0, lam[0:1], pk, Z1, half, U, beta, nfp, wsp, 1)
G.copy_(torch.randn(M, D, D, **f32)); G.copy_(G + G.transpose(1, 2))
grp, glp = torch.zeros(M, 28, **f32), torch.zeros(M, **f32)
for _ in range(3):
    lib.cell_bwd(G, S, Z0, half, U, beta, lam[0:1], pk, Go, grp, glp, 1, wsp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    lib.cell_bwd(G, S, Z0, half, U, beta, lam[0:1], pk, Go, grp, glp, 1, wsp)
e1.record(); torch.cuda.synchronize()
print(f"cell_bwd D={D} M={M} lean_bwd={os.environ.get('UGLAD_LEAN_BWD', '1')}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")
kb = (ctypes.c_ulonglong * 32)()
assert lib._dll.uglad_diag_kstamps(ctypes.cast(kb, ctypes.c_void_p)) == 0
k = np.array(list(kb), dtype=np.int64)
names = ["load U + spectrum", "phase A (rhoNN bwd)", "gemm 1", "gemm 2", "C o F", "gemm 3", "gemm 4", "G_out", "reductions"]
print("workgroup 0 phases (s_memtime ticks): " + "  ".join(f"{n} {int(k[i+1]-k[i])}" for i, n in enumerate(names)) + f"  total {int(k[9]-k[0])}")
asym = float((Go - Go.transpose(1, 2)).abs().max() / Go.abs().max())
print(f"G_out asymmetry (max |G - G^T| / max |G|): {asym:.2e};  checksum {float(Go.double().sum()):.6e} glam {float(glp.double().sum()):.6e} grho {float(grp.double().sum()):.6e}")
