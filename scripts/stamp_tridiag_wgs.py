#!/usr/bin/env python3
"""Diagnostic (GPU box): start / end clock and CU of every workgroup of one tridiag_kernel launch."""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.environ.get("UGLAD_DIAG_SO", os.path.join(ROOT, "scripts", "_build", "libuglad_diag.so"))  # scripts/dev_build.sh
from uglad_amd import _lib
from uglad_amd.utils.prepare_data import synthetic_covariance_batch
D = 128; M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lib = _lib.HipLib(so, require_gpu=True)
base = synthetic_covariance_batch(8, D, seed=5)
S = torch.from_numpy(np.tile(base, (M // 8 + 1, 1, 1))[:M]).cuda().contiguous()
Z = torch.eye(D, device="cuda").repeat(M, 1, 1).contiguous(); R = torch.empty_like(S)
lam = torch.full((1,), 0.5, device="cuda"); wsp = lib.workspace(M, D, S)
for _ in range(3):
    lib.tridiagonalize(S, Z, lam, R, wsp)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (3 * M))()
assert lib._dll.uglad_diag_twg(ctypes.cast(buf, ctypes.c_void_p), M) == 0
t = np.array(list(buf), dtype=np.int64).reshape(M, 3)
t0 = t[:, 0].min(); st = (t[:, 0] - t0) / 2400.0; en = (t[:, 1] - t0) / 2400.0  # us at ~2.4 GHz (s_memtime ticks = 100 MHz? see ratio below)
hw = t[:, 2] & 0xffffffff; xcc = t[:, 2] >> 32
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
cuid = xcc * 10000 + se * 1000 + sh * 100 + cu
print(f"M={M}: starts: min {st.min():.1f} median {np.median(st):.1f} max {st.max():.1f}; ends: median {np.median(en):.1f} max {en.max():.1f}; duration median {np.median(en - st):.1f} max {(en - st).max():.1f}  [clock ticks / 2400]")
u, c = np.unique(cuid, return_counts=True)
print(f"distinct CUs {len(u)}; workgroups per CU: " + ", ".join(f"{k}: {int((c == k).sum())} CUs" for k in sorted(set(c))))
late = st > 0.5 * np.median(en - st)
print(f"workgroups that start after half a workgroup duration (second round): {int(late.sum())}")
simd = (hw >> 4) & 0x3
per_cu = {}
for b in range(M):
    per_cu.setdefault(int(cuid[b]), []).append((b, int(simd[b])))
from collections import Counter
pat = Counter(tuple(sorted(s_ for _, s_ in v)) for v in per_cu.values())
print("SIMD of wave 0 of the workgroups sharing a CU (sorted tuple: number of CUs):", dict(pat))
ex = list(per_cu.items())[:6]
print("examples (CU: [(blockIdx, simd of wave 0)]):", ex)
