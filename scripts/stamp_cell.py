#!/usr/bin/env python3
"""Diagnostic (GPU box): phase cycles of cell_fwd / cell_bwd (workgroup 0) from a -DUGLAD_STAMPS build."""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "gpurun_out", "libuglad_diag.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DUGLAD_STAMPS", "-DUGLAD_THREADS=" + os.environ.get("UGLAD_THREADS", "512"),
                os.path.join(ROOT, "uglad_amd/csrc/glad_kernels.hip"), "-o", so], check=True)
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
Z0, Z1, half, U, G1 = (torch.empty(M, D, D, **f32) for _ in range(5))
beta, nfp = torch.empty(M, D, **f32), torch.empty(M, **f32)
lam, lam_in = torch.empty(2, **f32), torch.empty(2, 2, **f32)
wsp = lib.workspace(M, D, S)
lib.init_theta(S, pk, 0, Z0, wsp); lib.lambda_init(pk, 1.0, lam[0:1], lam_in[0])
G0 = torch.randn(M, D, D, **f32); G0 = (G0 + G0.transpose(1, 2)).contiguous()
grp, glp = torch.zeros(M, 28, **f32), torch.empty(M, **f32)
buf = (ctypes.c_ulonglong * 32)()
def stamps():
    torch.cuda.synchronize()
    assert lib._dll.uglad_diag_kstamps(ctypes.cast(buf, ctypes.c_void_p)) == 0
    return np.array(list(buf), dtype=np.int64)
_lib._SIGS["uglad_diag_tstamps"] = ([ctypes.c_void_p, ctypes.c_int], ctypes.c_int)
tb = (ctypes.c_ulonglong * 4)()
lib._dll.uglad_diag_tstamps(ctypes.cast(tb, ctypes.c_void_p), 1)
lib.cell_fwd(S, Z0, lam[0:1], pk, Z1, half, U, beta, nfp, wsp, 1)
torch.cuda.synchronize()
lib._dll.uglad_diag_tstamps(ctypes.cast(tb, ctypes.c_void_p), 1)
print(f"tridiag D={D} M={M} (workgroup 0, shader cycles over all {D - 2} steps): reflector chain + barrier {tb[0]}  sweep + barrier {tb[1]}  prologue (load + set-up) {tb[2]}")
for _ in range(2):
    lib.cell_fwd(S, Z0, lam[0:1], pk, Z1, half, U, beta, nfp, wsp, 1)
s = stamps()
print(f"cell_fwd D={D} (workgroup 0, shader cycles): load {s[16]-0 if False else 0}  solver {s[17]-s[16]}  phi+W {s[18]-s[17]}  gemm {s[19]-s[18]}  epilogue {s[20]-s[19]} (tiles->LDS {s[21]-s[19]}, rhoNN {s[22]-s[21]}, norm + copy-out {s[20]-s[22]})  total {s[20]-s[16]}")
for _ in range(2):
    lib.cell_bwd(G0, S, Z0, half, U, beta, lam[0:1], pk, G1, grp, glp, 1)
s = stamps()
names = ["load U+spectrum", "phase A (rhoNN bwd)", "gemm1 G_half U + store", "gemm2 U^T T1", "epilogue C o F", "gemm3 + store", "gemm4", "G_out via LDS", "reductions"]
print(f"cell_bwd D={D}: " + "  ".join(f"{n} {s[i+1]-s[i]}" for i, n in enumerate(names[:9])) + f"  total {s[9]-s[0]}")
