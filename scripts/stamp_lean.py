#!/usr/bin/env python3
"""Diagnostic (GPU box): phase cycles of the LDS-lean forward cell (eig_lean.h) from a -DUGLAD_STAMPS build made by
scripts/dev_build.sh (scripts/_build/libuglad_diag.so).  Prints workgroups 0..3 of one launch over M matrices."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.environ.get("UGLAD_DIAG_SO", os.path.join(ROOT, "scripts", "_build", "libuglad_diag.so"))
from uglad_amd import _lib
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
_lib._SIGS["uglad_diag_kstamps"] = ([ctypes.c_void_p], ctypes.c_int)
_lib._SIGS["uglad_diag_lstamps"] = ([ctypes.c_void_p], ctypes.c_int)
lib = _lib.HipLib(so, require_gpu=True)
from uglad_amd.utils.prepare_data import synthetic_covariance_batch
base = synthetic_covariance_batch(8, D, seed=5)
S = torch.from_numpy(np.tile(base, (M // 8 + 1, 1, 1))[:M]).cuda().contiguous()
pz = np.load(os.path.join(ROOT, "tests", "golden", "params_trained.npz"))
pk = torch.tensor(np.concatenate([pz[k].ravel() for k in pz.files]), dtype=torch.float32, device="cuda")
f32 = dict(dtype=torch.float32, device="cuda")
Z0, Z1, half, U = (torch.empty(M, D, D, **f32) for _ in range(4))
beta, nfp = torch.empty(M, D, **f32), torch.empty(M, **f32)
lam, lam_in = torch.empty(2, **f32), torch.empty(2, 2, **f32)
wsp = lib.workspace(M, D, S)
lib.init_theta(S, pk, 0, Z0, wsp); lib.lambda_init(pk, 1.0, lam[0:1], lam_in[0])
for _ in range(3):
    lib.cell_fwd(S, Z0, lam[0:1], pk, Z1, half, U, beta, nfp, wsp, 1)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (4 * 96))()
assert lib._dll.uglad_diag_lstamps(ctypes.cast(buf, ctypes.c_void_p)) == 0
s = np.array(list(buf), dtype=np.int64).reshape(4, 96)
kb = (ctypes.c_ulonglong * 32)()
assert lib._dll.uglad_diag_kstamps(ctypes.cast(kb, ctypes.c_void_p)) == 0
k = np.array(list(kb), dtype=np.int64)
def span(a, b, wg):
    return int(s[wg, b] - s[wg, a]) if s[wg, a] > 0 and s[wg, b] > 0 else -1
for wg in range(4):
    print(f"--- lean cell_fwd D={D} M={M} workgroup {wg} (shader cycles)")
    print(f"  divide & conquer {span(1, 40, wg)}   back-transform {span(40, 41, wg)}   solver total {span(0, 41, wg)}")
    lvl, h = 1, 2
    while h < D:
        b = 2 + 5 * lvl
        ev = s[wg, 80 + lvl]
        print(f"    level {lvl} (merge to {2*h:3d}): sort/perturb {span(b, b+1, wg):7d}  secular {span(b+1, b+2, wg):7d}  zhat {span(b+2, b+3, wg):7d}  gemm {span(b+3, b+4, wg):7d}"
              f"   evals/root mean {float(ev >> 32) / D:.2f} max {int(ev & 0xffffffff)}")
        lvl += 1; h *= 2
    nblk = (D - 2 + 15) // 16
    print(f"    Gram {span(42, 44, wg)}  T factors {span(44, 45, wg)}")
    prev = 45
    parts = []
    for b in range(nblk - 1, -1, -1):
        parts.append(f"{b}: {span(prev, 46 + b, wg)}")
        prev = 46 + b
    print("    blocks of 16 reflectors (stage + Y + TY + update): " + "  ".join(parts))
print(f"workgroup 0 kernel phases: solver {k[17]-k[16]}  phi+U out {k[18]-k[17]}  theta_half gemm {k[19]-k[18]}  tiles->LDS {k[21]-k[19]}  half out + rhoNN {k[22]-k[21]}  norm + copy-out {k[20]-k[22]}  total {k[20]-k[16]}")

# per-workgroup lifetimes of the launch (100 MHz wall clock): how the 1024 workgroups pack onto the CUs
wb = (ctypes.c_ulonglong * (3 * M))()
_lib._SIGS  # (signature not registered: call through the raw handle)
assert lib._dll.uglad_diag_cwg(ctypes.cast(wb, ctypes.c_void_p), M) == 0
w = np.array(list(wb), dtype=np.int64).reshape(M, 3)
t0 = w[:, 0].min()
st, en = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0  # microseconds
print(f"workgroup lifetimes (us): start min/median/max {st.min():.1f}/{np.median(st):.1f}/{st.max():.1f}; end median/max {np.median(en):.1f}/{en.max():.1f}; "
      f"duration min/median/max {(en-st).min():.1f}/{np.median(en-st):.1f}/{(en-st).max():.1f}")
cu = (w[:, 2] >> 32) * 100000 + ((w[:, 2] & 0xffffffff) >> 8 & 0xf) * 1000 + ((w[:, 2] & 0xffffffff) >> 13 & 0x7) * 100 + ((w[:, 2] & 0xffffffff) >> 12 & 1) * 10
ids, cnt = np.unique(w[:, 2] >> 0 & ~np.int64(0xff), return_counts=True)
print(f"distinct (xcc, hw_id without wave/simd bits) {len(ids)}; workgroups per id: min {cnt.min()} max {cnt.max()}")
order = np.argsort(st)
print("first-round starts (us):", np.round(st[order][:8], 1), "... starts of workgroups 512..519:", np.round(np.sort(st)[512:520], 1))
late = st > 0.25 * en.max()
print(f"{late.sum()} workgroups started after {0.25*en.max():.0f} us; durations first round median {np.median((en-st)[~late]):.1f}, later rounds median {np.median((en-st)[late]):.1f}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    lib.cell_fwd_stage2(S, Z0, lam[0:1], pk, Z1, half, U, beta, nfp, wsp, 1)
e0.record()
for _ in range(30):
    lib.cell_fwd_stage2(S, Z0, lam[0:1], pk, Z1, half, U, beta, nfp, wsp, 1)
e1.record(); torch.cuda.synchronize()
print(f"second stage alone (30 launches, HIP events): {e0.elapsed_time(e1) / 30 * 1e3:.1f} us per launch   [{so}]")

# stamps inside the secular solver (workgroup 0, thread 0): row = poles per lane (NP), columns: entry, poles loaded, test evaluation + origin, starting point, end; evaluations
try:
    sb = (ctypes.c_ulonglong * (16 * 8))()
    assert lib._dll.uglad_diag_sec(ctypes.cast(sb, ctypes.c_void_p)) == 0
    q = np.array(list(sb), dtype=np.int64).reshape(16, 8)
    for npl in range(16):
        if q[npl, 0] > 0 and q[npl, 4] > q[npl, 0]:
            print(f"secular solver, {npl if npl < 15 else '16+'} poles per lane (thread 0's last call): pole loads {q[npl,1]-q[npl,0]}  test evaluation + origin {q[npl,2]-q[npl,1]}  "
                  f"starting point {q[npl,3]-q[npl,2]}  iteration {q[npl,4]-q[npl,3]} ({q[npl,5]} evaluations)  total {q[npl,4]-q[npl,0]}")
except AttributeError:
    pass
