#!/usr/bin/env python3
"""Diagnostic (GPU box): what does workgroup co-residency buy the stage-2 solver?  Times uglad_symeig on M matrices of
order D (LDS-light for D <= 64: several workgroups per CU) with and without a dummy LDS allocation that forces one
workgroup per CU (-DUGLAD_LDS_PAD=bytes)."""
import ctypes, os, subprocess, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uglad_amd.utils.prepare_data import synthetic_covariance_batch
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
base = synthetic_covariance_batch(8, D, seed=5)
S = torch.from_numpy(np.tile(base, (M // 8 + 1, 1, 1))[:M]).cuda()
A = (S / 0.06 - torch.diag_embed(1.0 / (torch.diagonal(S, dim1=1, dim2=2) + 1.0))).contiguous()
U = torch.empty_like(A); beta = torch.empty(M, D, device="cuda")
vp = lambda t: ctypes.c_void_p(t.data_ptr())
for pad in (0, 60000, 100000):
    so = os.path.join(ROOT, "gpurun_out", f"libuglad_pad{pad}.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-DUGLAD_LDS_PAD={pad}",
                    os.path.join(ROOT, "uglad_amd/csrc/glad_kernels.hip"), "-o", so], check=True)
    dll = ctypes.CDLL(so)
    wsp = torch.empty(dll.uglad_workspace_floats(M, D), device="cuda")
    for _ in range(2):
        dll.uglad_symeig(vp(A), vp(U), vp(beta), vp(wsp), M, D, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        dll.uglad_symeig(vp(A), vp(U), vp(beta), vp(wsp), M, D, None)
    torch.cuda.synchronize()
    print(f"D={D} M={M} LDS pad {pad:6d} B: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per uglad_symeig (tridiag + stage 2)", flush=True)
