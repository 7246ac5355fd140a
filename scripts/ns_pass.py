#!/usr/bin/env python3
"""A few training passes on the matrix-iteration path, for rocprofv3: python scripts/ns_pass.py [D] [B] [L] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import uglad_amd  # noqa: E402
from uglad_amd.utils.prepare_data import synthetic_covariance_batch  # noqa: E402

D, B, L, reps = (int(a) for a in (sys.argv[1:5] + ["512", "1", "15", "3"][len(sys.argv) - 1:]))
S = torch.from_numpy(synthetic_covariance_batch(B, D, seed=D)).cuda()
torch.manual_seed(0)
model = uglad_amd.GladParams(1.0, device="cuda")
for _ in range(reps):
    model.zero_grad()
    theta, loss = uglad_amd.forward_uGLAD(S, model, L=L)
    loss.backward()
torch.cuda.synchronize()
print("loss", loss.item())
