#!/usr/bin/env python3
"""Timing of the matrix-iteration path (csrc/wide_ns.h) against the spectral path where both exist: one training pass (forward + loss +
backward), L = 15, per (D, batch).  gpurun: python scripts/ns_probe.py > gpurun_out/ns_probe.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import uglad_amd  # noqa: E402
from uglad_amd import _lib  # noqa: E402
from uglad_amd.utils.prepare_data import synthetic_covariance_batch  # noqa: E402


def one(D, B, L, forced, reps=8):
    lib = _lib.get_lib()
    lib.set_matrix_iteration(1 if forced else 0)  # (0: the spectral path wherever it exists -- the automatic choice would pick per shape)
    try:
        S = torch.from_numpy(synthetic_covariance_batch(B, D, seed=D)).cuda()
        torch.manual_seed(0)
        model = uglad_amd.GladParams(1.0, device="cuda")

        def step(train=True):
            if train:
                model.zero_grad()
                theta, loss = uglad_amd.forward_uGLAD(S, model, L=L)
                loss.backward()
            else:
                with torch.no_grad():
                    uglad_amd.forward_uGLAD(S, model, L=L)

        out = []
        for train in (False, True):
            step(train)
            step(train)
            torch.cuda.synchronize()
            # the MEDIAN of single passes: an average over the loop sometimes carries an allocator stall of tens of ms when a size first misses
            # torch's pool (profiles/r04_cov_probe.txt) -- round 4's first table had 13 ms "forward-only" next to 9 ms for the training pass
            ts = []
            for _ in range(reps):
                t = time.perf_counter()
                step(train)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t)
            ts.sort()
            out.append(ts[len(ts) // 2] * 1e3)
        return out
    finally:
        lib.set_matrix_iteration(-1)


def main():
    L = 15
    print(f"# ms per pass, L = {L}: forward only / forward + loss + backward")
    for D, B in ((128, 8), (256, 1), (256, 8), (256, 64), (320, 1), (384, 1), (512, 1), (512, 8), (768, 1), (1024, 1), (2048, 1)):
        row = f"D={D:4d} B={B:3d}"
        if D <= 256:
            f, t = one(D, B, L, False)
            row += f"   spectral {f:8.2f} / {t:8.2f}"
        f, t = one(D, B, L, True)
        row += f"   matrix iteration {f:8.2f} / {t:8.2f}"
        print(row, flush=True)


if __name__ == "__main__":
    main()
