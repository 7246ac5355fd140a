#!/usr/bin/env python3
"""Diagnostic (GPU box): for every cell / regime golden, per parameter tensor, the three distances
    kernel <-> fp64 oracle (oracle/glad_exact.py, the reference's function in exact arithmetic, mode ns10),
    reference <-> fp64 oracle (= tests/golden/grad_noise_floor.json, the reference's own fp32 noise),
    kernel <-> reference,
written to gpurun_out/grad_vs_fp64.json (+ a table on stdout).  This is where the fixed per-class tolerances of
tests/test_gpu_parity.py::FP64_GRAD_TOL come from (profiles/r04_grad_vs_fp64.txt)."""
import glob
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uglad_amd  # noqa: E402
from oracle import glad_exact as ex  # noqa: E402  (the checker)
from uglad_amd import _lib  # noqa: E402


def relF(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    lib = _lib.get_lib()
    noise = json.load(open(os.path.join(ROOT, "tests", "golden", "grad_noise_floor.json")))
    out = {}
    paths = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "cell_*.npz")) + glob.glob(os.path.join(ROOT, "tests", "golden", "regime_*.npz")))
    print(f"library: {lib.path}")
    print(f"{'golden':42s} {'Theta k-64':>10s} {'Theta r-64':>10s} | worst tensor: kernel-fp64  ref-fp64(noise)  kernel-ref | max ratio k64/noise")
    for path in paths:
        g = np.load(path)
        name = os.path.basename(path)[:-4]
        D = g["S"].shape[-1]
        if D > lib.max_dim:
            continue
        m = uglad_amd.GladParams(1.0, device="cuda")
        m.load_state_dict({k: torch.from_numpy(np.array(g["param." + k])) for k in ex.PARAM_KEYS})
        kw = {}
        loss_S = g["loss_S"] if "loss_S" in g else None
        struct = g["struct"] if "struct" in g else None
        if loss_S is not None:
            kw["loss_Sb"] = torch.from_numpy(loss_S).cuda()
        if struct is not None:
            kw["struct_theta"] = torch.from_numpy(struct).cuda()
        L, diag = int(g["L"]), int(g["INIT_DIAG"])
        th, ls = uglad_amd.forward_uGLAD(torch.from_numpy(g["S"]).cuda(), m, L=L, INIT_DIAG=diag, **kw)
        if not torch.isfinite(ls):
            print(f"{name:42s} loss not finite (reference: {float(g['loss'])})")
            continue
        ls.backward()
        sd = dict(m.named_parameters())
        p = ex.params64(g, "param.")
        t64, tr = ex.glad_forward(g["S"], p, L, diag, loss_S=loss_S, struct=struct, mode="ns10")
        g64 = ex.glad_backward(g["S"], p, L, tr, diag, loss_S=loss_S, struct=struct, mode="ns10")
        thn = th.detach().cpu().numpy()
        rec = {"D": int(D), "L": L,
               "theta_kernel_fp64": max(relF(thn[i], t64[i]) for i in range(thn.shape[0])),
               "theta_ref_fp64": max(relF(g["theta_L"][i], t64[i]) for i in range(thn.shape[0])),
               "theta_kernel_ref": max(relF(thn[i], g["theta_L"][i]) for i in range(thn.shape[0])),
               "grads": {}}
        for k in ex.PARAM_KEYS:
            got = sd[k].grad.cpu().numpy()
            rec["grads"][k] = {"kernel_fp64": relF(got, g64[k]), "ref_fp64": relF(g["grad." + k], g64[k]),
                               "kernel_ref": relF(got, g["grad." + k])}
        out[name] = rec
        w = max(rec["grads"], key=lambda k: rec["grads"][k]["kernel_fp64"])
        ratio = max(v["kernel_fp64"] / max(v["ref_fp64"], 1e-12) for v in rec["grads"].values())
        gw = rec["grads"][w]
        print(f"{name:42s} {rec['theta_kernel_fp64']:10.2e} {rec['theta_ref_fp64']:10.2e} | {w:18s} {gw['kernel_fp64']:.2e}  {gw['ref_fp64']:.2e}  "
              f"{gw['kernel_ref']:.2e} | {ratio:.2f}", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "grad_vs_fp64.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
