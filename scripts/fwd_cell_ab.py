"""Same-box A/B of the forward cell (tridiag + lean launch, M = 1024, D = 128) between two development builds under scripts/_build
(scripts/dev_build.sh <name>, or a single-translation-unit build of another tree): python scripts/fwd_cell_ab.py <name_a> <name_b>"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
from uglad_amd import _lib
from uglad_amd.utils.prepare_data import synthetic_covariance_batch
D, M = 128, 1024
libs = {n: _lib.HipLib(os.path.join(ROOT, "scripts", "_build", f"libuglad_{n}.so"), require_gpu=True) for n in (sys.argv[1:3] if len(sys.argv) > 2 else ("oldcopy", "cur"))}
base = synthetic_covariance_batch(8, D, seed=5)
S = torch.from_numpy(np.tile(base, (M // 8 + 1, 1, 1))[:M]).cuda().contiguous()
pz = np.load(os.path.join(ROOT, "tests", "golden", "params_trained.npz"))
pk = torch.tensor(np.concatenate([pz[k].ravel() for k in pz.files]), dtype=torch.float32, device="cuda")
f32 = dict(dtype=torch.float32, device="cuda")
Z0, Z1, half, U = (torch.empty(M, D, D, **f32) for _ in range(4))
beta, nfp = torch.empty(M, D, **f32), torch.empty(M, **f32)
lam, lam_in = torch.empty(2, **f32), torch.empty(2, 2, **f32)
res = {n: [] for n in libs}
for rnd in range(6):
    for n, lib in libs.items():
        wsp = lib.workspace(M, D, S)
        lib.init_theta(S, pk, 0, Z0, wsp); lib.lambda_init(pk, 1.0, lam[0:1], lam_in[0])
        for _ in range(3):
            lib.cell_fwd(S, Z0, lam[0:1], pk, Z1, half, U, beta, nfp, wsp, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            lib.cell_fwd(S, Z0, lam[0:1], pk, Z1, half, U, beta, nfp, wsp, 1)
        e1.record(); torch.cuda.synchronize()
        res[n].append(e0.elapsed_time(e1) / 30)
for n in res:
    print(n, "forward cell (tridiag + lean) ms per launch:", " ".join(f"{v:.4f}" for v in res[n]), " median", f"{np.median(res[n]):.4f}")
