#!/usr/bin/env python3
"""Development probe (GPU box): per-phase hardware counters of the LDS-lean forward cell.  A -DUGLAD_PHASE_EXIT build
(EXTRA=-DUGLAD_PHASE_EXIT bash scripts/dev_build.sh exit) ends every wave at a chosen phase boundary; this script launches the cell once
per boundary, in execution order, and `rocprofv3 --pmc ... -- python scripts/phase_exit_probe.py run` records the counters per launch;
`python scripts/phase_exit_probe.py report <dir>` prints the differences between successive cuts = the phases' own counts."""
import csv
import ctypes
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# boundaries in execution order (eig_lean.h stamps; 100 + k: the kernel's own KSTAMP(k)); name = the phase that ENDS there
CUTS = [(27, "local merges up to 32"), (28, "merge to 64: sort/perturb"), (29, "merge to 64: secular"), (30, "merge to 64: zhat"),
        (31, "merge to 64: eigenvector product"), (33, "merge to 128: sort/perturb"), (34, "merge to 128: secular"),
        (35, "merge to 128: zhat"), (36, "merge to 128: eigenvector product"), (42, "back-transform: prologue"),
        (44, "back-transform: Gram"), (45, "back-transform: T factors")] + \
       [(46 + b, f"back-transform: reflector block {b}") for b in range(7, -1, -1)] + \
       [(117, "solver epilogue"), (118, "spectrum + U out"), (119, "U psi U^T product"), (121, "tiles -> LDS"),
        (122, "theta_half out + rhoNN + threshold"), (-1, "norm + Z out (to the end)")]


def run():
    import numpy as np
    import torch
    from uglad_amd import _lib
    from uglad_amd.utils.prepare_data import synthetic_covariance_batch
    D, M = 128, 1024
    so = os.environ.get("UGLAD_DIAG_SO", os.path.join(ROOT, "scripts", "_build", "libuglad_exit.so"))
    _lib._SIGS["uglad_diag_set_exit"] = ([ctypes.c_int], ctypes.c_int)
    lib = _lib.HipLib(so, require_gpu=True)
    base = synthetic_covariance_batch(8, D, seed=5)
    S = torch.from_numpy(np.tile(base, (M // 8 + 1, 1, 1))[:M]).cuda().contiguous()
    pz = np.load(os.path.join(ROOT, "tests", "golden", "params_trained.npz"))
    pk = torch.tensor(np.concatenate([pz[k].ravel() for k in pz.files]), dtype=torch.float32, device="cuda")
    f32 = dict(dtype=torch.float32, device="cuda")
    Z0, Z1, half, U = (torch.empty(M, D, D, **f32) for _ in range(4))
    beta, nfp = torch.empty(M, D, **f32), torch.empty(M, **f32)
    lam, lam_in = torch.empty(2, **f32), torch.empty(2, 2, **f32)
    wsp = lib.workspace(M, D, S)
    assert lib._dll.uglad_diag_set_exit(-1) == 0
    lib.init_theta(S, pk, 0, Z0, wsp)
    lib.lambda_init(pk, 1.0, lam[0:1], lam_in[0])
    for _ in range(2):  # two whole launches first (warm-up), then one per cut
        lib.cell_fwd(S, Z0, lam[0:1], pk, Z1, half, U, beta, nfp, wsp, 1)
    torch.cuda.synchronize()
    for at, _name in CUTS:
        assert lib._dll.uglad_diag_set_exit(at) == 0
        lib.cell_fwd(S, Z0, lam[0:1], pk, Z1, half, U, beta, nfp, wsp, 1)
        torch.cuda.synchronize()
    print("launched", len(CUTS), "cuts")


def report(d):
    rows = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "cell_fwd_lean_kernel" not in r["Kernel_Name"]:
                continue
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)[-len(CUTS):]
    names = sorted({c for i in ids for c in rows[i]})
    print(f"{'phase':44s}" + "".join(f"{c[3:]:>22s}" for c in names))
    prev = {c: 0.0 for c in names}
    for (at, name), i in zip(CUTS, ids):
        cur = rows[i]
        print(f"{name:44s}" + "".join(f"{cur.get(c, 0.0) - prev[c]:22.0f}" for c in names))
        prev = {c: cur.get(c, 0.0) for c in names}
    print(f"{'whole kernel':44s}" + "".join(f"{prev[c]:22.0f}" for c in names))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        run()
