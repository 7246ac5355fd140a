#!/usr/bin/env python3
"""Diagnostic (GPU box): where does the eigensolver spend its cycles?  Builds a -DUGLAD_STAMPS copy of the library
(never the shipped one), runs the solver on M matrices and prints the median shader-clock cycles per phase."""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "gpurun_out", "libuglad_diag.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DUGLAD_STAMPS", "-DUGLAD_THREADS=" + os.environ.get("UGLAD_THREADS", "512"),
                os.path.join(ROOT, "uglad_amd/csrc/glad_kernels.hip"), "-o", so], check=True)
dll = ctypes.CDLL(so)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M = int(sys.argv[2]) if len(sys.argv) > 2 else 256
from uglad_amd.utils.prepare_data import synthetic_covariance_batch
base = synthetic_covariance_batch(8, D, seed=5)
S = torch.from_numpy(np.tile(base, (M // 8 + 1, 1, 1))[:M]).cuda()
A = (S / 0.06 - torch.diag_embed(1.0 / (torch.diagonal(S, dim1=1, dim2=2) + 1.0))).contiguous()
U = torch.empty_like(A); beta = torch.empty(M, D, device="cuda")
st = torch.zeros(M, 96, dtype=torch.int64, device="cuda")
wsp = torch.empty(dll.uglad_workspace_floats(M, D), device="cuda")
vp = lambda t: ctypes.c_void_p(t.data_ptr())
for _ in range(2):
    rc = dll.uglad_symeig_stamps(vp(A), vp(U), vp(beta), vp(wsp), M, D, vp(st), None)
    torch.cuda.synchronize()
assert rc == 0
s = st.cpu().numpy().astype(np.int64)
def span(a, b):
    v = s[:, b] - s[:, a]
    v = v[(s[:, a] > 0) & (s[:, b] > 0)]
    return float(np.median(v)) if len(v) else float("nan")
print(f"D={D} M={M}  (median shader cycles per workgroup)")
print("  (tridiagonalisation runs in its own kernel: see rocprofv3 kernel stats)")
print(f"  divide & conquer    {span(1, 40):12.0f}")
lvl = 0
h = 1
while h < D:
    b = 2 + 5 * lvl
    ev = s[:, 80 + lvl]
    print(f"    level {lvl} (merge to {2*h:3d}): sort/perturb {span(b, b+1):9.0f}  secular {span(b+1, b+2):9.0f}  vectors {span(b+2, b+3):9.0f}  gemm {span(b+3, b+4):9.0f}"
          f"   secular evaluations/root: mean {float(np.mean(ev >> 32)) / D:.2f}, max over roots {float(np.mean(ev & 0xffffffff)):.1f} (worst matrix {int((ev & 0xffffffff).max())})")
    lvl += 1; h *= 2
print(f"  back-transform      {span(40, 41):12.0f}")
print(f"    load reflectors {span(42, 43):9.0f}  Gram {span(43, 44):9.0f}  T factors {span(44, 45):9.0f}")
nblk = (D - 2 + 31) // 32
for b in range(nblk - 1, -1, -1):
    prev = 45 if b == nblk - 1 else 48 + 4 * (b + 1)
    print(f"    block {b}: Y=VQ {span(prev, 46+4*b):9.0f}  Y=TY {span(46+4*b, 47+4*b):9.0f}  update {span(47+4*b, 48+4*b):9.0f}")
print(f"  total               {span(0, 41):12.0f}")
rec = (U * beta[:, None, :]) @ U.transpose(1, 2)
print("  recon err", float(((rec - A).flatten(1).norm(dim=1) / A.flatten(1).norm(dim=1)).max()))
