#!/usr/bin/env python3
"""Where the matrix-iteration cell overtakes the spectral cell: ms per 15-step pass (forward only / training) on both paths over a grid
of (D, batch).  python scripts/ns_crossover.py > gpurun_out/ns_crossover.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ns_probe import one
print("# ms per pass, L = 15: forward only / training;  spectral | matrix iteration;  '<' marks the faster path per column")
for D in (64, 96, 128, 160, 192, 224, 256):
    for B in (1, 2, 4, 6, 8, 16):
        f0, t0 = one(D, B, 15, False, reps=5)
        f1, t1 = one(D, B, 15, True, reps=5)
        print(f"D={D:4d} B={B:3d}: {f0:7.2f} / {t0:7.2f} | {f1:7.2f} / {t1:7.2f}   fwd {'iter' if f1 < f0 else 'spec'}  train {'iter' if t1 < t0 else 'spec'}", flush=True)
