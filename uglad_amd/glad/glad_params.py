"""The 42 learnable floats of the GLAD cell -- same module tree and `state_dict` keys as the reference's
`GladParams` (uglad/glad/glad_params.py:6-95), so weights trained by the reference load unchanged:

    theta_init_offset (1,)                                  Theta_0 = (S + t I)^-1
    rho_l1.{0,2,4}.{weight,bias}   3->3->3->1 tanh,tanh,sigmoid   entrywise threshold rho_ij  (28 floats)
    lambda_f.{0,2}.{weight,bias}   2->3->1   tanh,sigmoid         lambda_{k+1} = Lambda(normF, lambda_k) (13 floats)

The HIP kernels evaluate both networks themselves from the packed vector returned by `packed()`; `eta_forward` and
`lambda_forward` below exist for API parity with the reference (plain torch ops, not used by `glad()`).
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch import Tensor

PARAM_KEYS = (
    "theta_init_offset",
    "rho_l1.0.weight", "rho_l1.0.bias", "rho_l1.2.weight", "rho_l1.2.bias", "rho_l1.4.weight", "rho_l1.4.bias",
    "lambda_f.0.weight", "lambda_f.0.bias", "lambda_f.2.weight", "lambda_f.2.bias",
)
NPARAM = 42


class GladParams(nn.Module):
    def __init__(self, theta_init_offset: float, nF: int = 3, H: int = 3, USE_CUDA: bool = False, device=None) -> None:
        super().__init__()
        if nF != 3 or H != 3:
            raise ValueError("the gfx950 kernels implement the reference's fixed nF=3, H=3 networks (main.py:381-383)")
        self.nF, self.H = nF, H
        self.theta_init_offset = nn.Parameter(torch.tensor([float(theta_init_offset)], dtype=torch.float32))
        # same construction order as the reference, so torch.manual_seed gives the same draw (glad_params.py:38-59)
        self.rho_l1 = nn.Sequential(nn.Linear(nF, H), nn.Tanh(), nn.Linear(H, H), nn.Tanh(), nn.Linear(H, 1), nn.Sigmoid())
        self.lambda_f = nn.Sequential(nn.Linear(2, H), nn.Tanh(), nn.Linear(H, 1), nn.Sigmoid())
        if device is not None:
            self.to(device)

    def packed(self) -> Tensor:
        """The (42,) vector the kernels read; differentiable w.r.t. the 11 parameters (include/uglad_hip.h layout)."""
        return torch.cat([p.reshape(-1) for p in self.parameters()])

    # ---- API parity with the reference (not on the hot path)
    def eta_forward(self, X: Tensor, S: Tensor, k: int, F3: Tensor = None) -> Tensor:
        feats = [X.reshape(X.shape[0], -1, 1), S.expand_as(X).reshape(X.shape[0], -1, 1)]
        if F3 is not None:
            feats.append(F3.reshape(X.shape[0], -1, 1))
        rho = self.rho_l1(torch.cat(feats, dim=-1)).reshape(X.shape)
        return torch.sign(X) * torch.clamp_min(torch.abs(X) - rho, 0.0)

    def lambda_forward(self, normF, prev_lambda, k: int = 0) -> Tensor:
        w = self.lambda_f[0].weight
        x = torch.tensor([float(normF), float(prev_lambda)], dtype=w.dtype, device=w.device)  # detached, as glad_params.py:94
        return self.lambda_f(x)
