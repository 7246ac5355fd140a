"""The unrolled GLAD cell -- host side.  Mirrors `uglad/glad/glad.py`: `glad()` (:74-151), `get_optimizers` (:11-36),
`get_frobenius_norm` (:60-71), `batch_matrix_sqrt` (:39-57, replaced: see below).

`glad()` is ONE autograd node.  Forward = Theta_0 init + L x {cell kernel, normF reduce, LambdaNN step} enqueued on the
current HIP stream with no device->host copy (the reference syncs to the host every step to build the LambdaNN input,
glad.py:147); backward = L reverse cell kernels + one finishing kernel producing the 42 parameter gradients.  What the
reference keeps alive through autograd (~22 D^2-tensors per step) shrinks to (Z_k, theta_half_k, U_k, beta_k, lambda_k).

`sqrt_mode` selects how (b^T b + 4/lam I)^(1/2) acts on the spectrum of b: "ns10" reproduces the reference's 10-step
Newton-Schulz forward and its 10-step approximate backward (torch_sqrtm.py) eigenvalue by eigenvalue -- the drop-in
default; "exact" is the true square root (what NS converges to).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor
from torch.optim import Adam, Optimizer

from .. import _lib
from ..dist import Collective, get_collective

DEFAULT_SQRT_MODE = "ns10"


# ------------------------------------------------------------------------------------------------ regime diagnostic
class UgladRegimeWarning(UserWarning):
    """A pass left the regime in which parity with the reference is validated (see `RegimeMonitor`)."""


class RegimeMonitor:
    """Collects, for every `glad()` / `glad_grouped()` pass run while it is active, the largest cond_2(b^T b + 4/lam I) over the
    batch and the L steps (the kernels' `cond_max` output, include/uglad_hip.h).  The reference evaluates the square root of that
    matrix with 10 Newton-Schulz steps (torch_sqrtm.py:13-29); they are an accurate square root -- and the reference's fp32 matrix
    arithmetic a function of the spectrum alone, which is what this package reproduces -- only while the number is moderate
    (hard part 1).  uGLAD's own min-max-normalised inputs stay below ~100 in short fits and reach ~1e3 when a fit runs to convergence; `_lib.get_lib().validated_cond` is
    the bound up to which reference-made goldens confirm the 1e-4 tolerance.  Nothing is copied to the host until `result()`."""

    # a pass on the matrix-iteration path reports the Gershgorin UPPER BOUND of the number (include/uglad_hip.h, uglad_cond_is_upper_bound):
    # 1.3 ... 1.4 x the true value on sample covariances, more on others.  Such passes are held to kUpperBoundSlack x the validated bound, so
    # that a regime the spectral path would accept does not warn just because the number is an overestimate (round 3 compared both with
    # the same threshold: spurious warnings for D > 128); what it costs is that a pass up to that factor outside may go unwarned.
    kUpperBoundSlack = 4.0

    def __init__(self):
        self._parts = []  # 0-dim device tensors, one per pass (list.append is atomic: passes of several host threads may report)
        self._bounds = []  # the same for passes that report an upper bound

    def report(self, cond_per_matrix: Tensor, upper_bound: bool = False) -> None:
        (self._bounds if upper_bound else self._parts).append(cond_per_matrix.max())

    @staticmethod
    def _max(parts) -> float:
        if not parts:
            return 0.0
        if parts[0].is_cuda:
            torch.cuda.synchronize(parts[0].device)
        return float(torch.stack([p.reshape(()) for p in parts]).max().item())

    def result(self) -> float:
        """Maximum over everything reported so far (synchronises with the device); 0.0 when nothing ran.  Exact condition numbers and
        upper bounds alike: an upper bound of the run's largest condition number."""
        return max(self._max(self._parts), self._max(self._bounds))

    def warn_if_outside(self, where: str, coll: Optional["Collective"] = None) -> float:
        import warnings

        exact, upper = self._max(self._parts), self._max(self._bounds)
        if coll is not None and coll.world_size > 1:  # sharded batch: every rank warns about the global maximum
            t = torch.tensor([-exact, -upper], dtype=torch.float32, device=_lib.device())
            t = coll.all_reduce_min(t)
            exact, upper = -float(t[0].item()), -float(t[1].item())
        bound = _lib.get_lib().validated_cond
        if exact > bound or upper > self.kUpperBoundSlack * bound:
            what = (f"reached {exact:.3g}" if exact > bound else
                    f"has the Gershgorin upper bound {upper:.3g} (matrix-iteration path; held to {self.kUpperBoundSlack:g} x the validated bound)")
            warnings.warn(
                f"{where}: cond(b^T b + 4/lambda I) {what} (validated up to {bound:.3g}).  Beyond that the reference's "
                "10-step Newton-Schulz square root (uglad/glad/torch_sqrtm.py) is far from converged and its fp32 matrix arithmetic is "
                "no longer reproduced to 1e-4.  Reference-made goldens confirm the tolerance up to that bound: the min-max-normalised fit "
                "that runs to convergence reaches 1.0e3, covariances of raw samples 7e2; the case at 4.4e3 does not hold it.  Check the "
                "scaling of the input.",
                UgladRegimeWarning, stacklevel=3)
        return max(exact, upper)


_monitors: list = []


class regime_monitor:
    """Context manager: `with regime_monitor() as mon: ...; mon.result()`.  Monitors nest; every active one is reported to."""

    def __enter__(self) -> RegimeMonitor:
        self.mon = RegimeMonitor()
        _monitors.append(self.mon)
        return self.mon

    def __exit__(self, *exc):
        _monitors.remove(self.mon)
        return False


def get_optimizers(model_glad, lr_glad: float = 0.002, use_optimizer: str = "adam") -> Optimizer:
    """Adam(lr, betas=(0.9, 0.999), eps=1e-8) on the 42 parameters (ref glad.py:11-36)."""
    if use_optimizer == "adam":
        return Adam(model_glad.parameters(), lr=lr_glad, betas=(0.9, 0.999), eps=1e-08)
    raise ValueError("Optimizer not found! Supported optimizers: ['adam']")


def get_frobenius_norm(A: Tensor, single: bool = False) -> Tensor:
    """||A||_F^2 of one matrix, or its mean over a batch (ref glad.py:60-71).  Plain torch; inside `glad()` the same
    quantity comes out of the cell kernel's epilogue."""
    return torch.sum(A**2) if single else torch.mean(torch.sum(A**2, dim=(1, 2)))


def batch_symeig(A: Tensor):
    """Batched symmetric eigendecomposition on the GPU, A = U diag(beta) U^T.  Stands where the reference has
    `batch_matrix_sqrt` (glad.py:39-57): the cell needs f(b) for symmetric b, and gets it from the spectrum."""
    lib = _lib.get_lib()
    A = A.contiguous()
    if A.dim() == 2:
        A = A[None]
    U = torch.empty_like(A)
    beta = torch.empty(A.shape[:2], dtype=A.dtype, device=A.device)
    lib.symeig(A, U, beta)
    return beta, U


class _GladUnrolled(torch.autograd.Function):
    @staticmethod
    def forward(ctx, S: Tensor, params: Tensor, L: int, init_diag: int, lambda_init: float, mode: int,
                coll: Collective, m_global: int):
        lib = _lib.get_lib()
        M, D, _ = S.shape
        dev = S.device
        train = ctx.needs_input_grad[1]
        f32 = dict(dtype=torch.float32, device=dev)
        params = params.detach().contiguous()
        lam = torch.empty(L + 1, **f32)
        lam_in = torch.empty(L + 1, 2, **f32)
        nf_partial = torch.empty(M, **f32)
        nf_sum = torch.empty(1, **f32)
        cond = torch.zeros(M, **f32) if _monitors else None  # regime diagnostic: only when somebody listens
        if train:
            Z = torch.empty(L + 1, M, D, D, **f32)
            half = torch.empty(L, M, D, D, **f32)
            U = torch.empty(L, M, D, D, **f32)
            beta = torch.empty(L, M, D, **f32)
        else:
            Z = torch.empty(2, M, D, D, **f32)
        wsp = lib.workspace(M, D, S)
        fused = type(coll) is Collective and m_global == M  # plain single-process run: nothing to exchange between steps
        native = None if fused else getattr(coll, "native_exchange", lambda: None)()
        if native is not None:
            # sharded over RCCL: the whole pass in one library call, ncclAllReduce of the per-step scalar issued from C on this stream
            lib.glad_forward_sharded(S, params, lambda_init, init_diag, L, Z, half if train else None, U if train else None,
                                     beta if train else None, lam, lam_in, nf_partial, nf_sum, wsp, mode, m_global, native,
                                     cond_max=cond)
            fused = True
        elif fused:
            # one library call enqueues the whole pass (no per-step Python between the launches)
            lib.glad_forward(S, params, lambda_init, init_diag, L, Z, half if train else None, U if train else None,
                             beta if train else None, lam, lam_in, nf_partial, nf_sum, wsp, mode, cond_max=cond)
        else:
            lib.init_theta(S, params, init_diag, Z[0], wsp)
            lib.lambda_init(params, lambda_init, lam[0:1], lam_in[0])
        inv_m = 1.0 / float(m_global)
        for k in range(0 if fused else L):
            zi, zo = (Z[k], Z[k + 1]) if train else (Z[k & 1], Z[(k + 1) & 1])
            lib.cell_fwd(S, zi, lam[k:k + 1], params, zo, half[k] if train else None, U[k] if train else None,
                         beta[k] if train else None, nf_partial, wsp, mode, cond_max=cond)
            lib.sum_partials(nf_partial, nf_sum)
            coll.all_reduce_sum(nf_sum)
            lib.lambda_step(nf_sum, inv_m, lam[k:k + 1], params, lam[k + 1:k + 2], lam_in[k + 1])
        out = (Z[L] if train else Z[L & 1]).clone()
        if cond is not None:
            ub = lib.cond_is_upper_bound(M, D, train, mode)
            for mon in list(_monitors):
                mon.report(cond, upper_bound=ub)
        if train:
            ctx.save_for_backward(S, params, Z, half, U, beta, lam, lam_in)
            ctx.cfg = (L, init_diag, mode)
        ctx.mark_non_differentiable(lam)
        return out, lam

    @staticmethod
    def backward(ctx, G: Tensor, _glam):
        lib = _lib.get_lib()
        S, params, Z, half, U, beta, lam, lam_in = ctx.saved_tensors
        L, init_diag, mode = ctx.cfg
        M, D, _ = S.shape
        f32 = dict(dtype=torch.float32, device=S.device)
        bufs = (torch.empty(M, D, D, **f32), torch.empty(M, D, D, **f32))
        grad_rho_partial = torch.empty(M, _lib.NRHO, **f32)
        glam_partial = torch.empty(L, M, **f32)
        gt_partial = torch.empty(M, **f32)
        grad = torch.empty(_lib.NPARAM, **f32)
        wsp = lib.workspace(M, D, S)  # read by the beyond-LDS instantiations (D > 128) only
        # the parameter gradients are sums over the LOCAL matrices; a sharded caller all-reduces them (uglad_amd/dist.py)
        lib.glad_backward(G.contiguous(), S, params, init_diag, L, Z, half, U, beta, lam, lam_in, bufs[0], bufs[1],
                          grad_rho_partial, glam_partial, gt_partial, grad, wsp, mode)
        return None, grad, None, None, None, None, None, None


def glad(
    Sb: Tensor,
    model,
    lambda_init: float = 1,
    L: int = 15,
    INIT_DIAG: int = 0,
    USE_CUDA: bool = True,
    sqrt_mode: Optional[str] = None,
    collective: Optional[Collective] = None,
    global_batch: Optional[int] = None,
    return_lambdas: bool = False,
):
    """Unrolled alternating-minimisation for graphical lasso; signature of the reference's `glad.glad` (glad.py:74-81)
    plus additive keywords.  Sb: (B, D, D) or (D, D) sample covariances -> Theta (B, D, D) fp32 on the GPU.

    When the batch is sharded over ranks, `collective` carries the per-step SUM of the batch-wide norm and
    `global_batch` is the number of matrices over all ranks (the divisor of get_frobenius_norm's batch mean).
    """
    if sqrt_mode is None:
        sqrt_mode = DEFAULT_SQRT_MODE
    if sqrt_mode not in _lib.SQRT_MODES:
        raise ValueError(f"sqrt_mode must be one of {sorted(_lib.SQRT_MODES)}")
    if INIT_DIAG not in (0, 1):
        raise ValueError("INIT_DIAG must be 0 or 1")
    if Sb.dim() == 2:
        Sb = Sb.reshape(1, Sb.shape[0], Sb.shape[1])
    params = model.packed()
    Sb = Sb.detach().to(device=params.device, dtype=torch.float32).contiguous()
    coll = collective if collective is not None else get_collective()
    if global_batch is None and coll.world_size > 1:
        # (the shards of a batch that does not divide evenly differ in size: the divisor of the batch mean cannot be guessed locally)
        raise ValueError("glad(): a sharded batch needs global_batch = the number of matrices over all ranks")
    m_global = int(global_batch) if global_batch is not None else Sb.shape[0]
    theta, lam = _GladUnrolled.apply(Sb, params, int(L), int(INIT_DIAG), float(lambda_init), _lib.SQRT_MODES[sqrt_mode],
                                     coll, m_global)
    return (theta, lam) if return_lambdas else theta


# ------------------------------------------------------------------------------------------------ grouped passes
class _GladGrouped(torch.autograd.Function):
    """G independent GLAD problems in ONE batch: group g owns matrices [g M/G, (g+1) M/G), params[g] (42 floats) and its
    own lambda sequence (uglad_glad_forward_grouped / uglad_glad_backward_grouped)."""

    @staticmethod
    def forward(ctx, S: Tensor, params: Tensor, L: int, init_diag: int, lambda_init: float, mode: int):
        lib = _lib.get_lib()
        M, D, _ = S.shape
        G = params.shape[0]
        train = ctx.needs_input_grad[1]
        f32 = dict(dtype=torch.float32, device=S.device)
        params = params.detach().contiguous()
        lam = torch.empty(L + 1, G, **f32)
        lam_in = torch.empty(L + 1, G, 2, **f32)
        nf_partial = torch.empty(M, **f32)
        nf_sum = torch.empty(G, **f32)
        if train:
            Z = torch.empty(L + 1, M, D, D, **f32)
            half = torch.empty(L, M, D, D, **f32)
            U = torch.empty(L, M, D, D, **f32)
            beta = torch.empty(L, M, D, **f32)
        else:
            Z = torch.empty(2, M, D, D, **f32)
            half = U = beta = None
        wsp = lib.workspace(M, D, S)
        cond = torch.zeros(M, **f32) if _monitors else None
        lib.glad_forward(S, params, lambda_init, init_diag, L, Z, half, U, beta, lam, lam_in, nf_partial, nf_sum, wsp, mode,
                         groups=G, cond_max=cond)
        if cond is not None:
            ub = lib.cond_is_upper_bound(M, D, train, mode)
            for mon in list(_monitors):
                mon.report(cond, upper_bound=ub)
        out = (Z[L] if train else Z[L & 1]).clone()
        if train:
            ctx.save_for_backward(S, params, Z, half, U, beta, lam, lam_in)
            ctx.cfg = (L, init_diag, mode)
        return out

    @staticmethod
    def backward(ctx, Gout: Tensor):
        lib = _lib.get_lib()
        S, params, Z, half, U, beta, lam, lam_in = ctx.saved_tensors
        L, init_diag, mode = ctx.cfg
        M, D, _ = S.shape
        G = params.shape[0]
        f32 = dict(dtype=torch.float32, device=S.device)
        bufs = (torch.empty(M, D, D, **f32), torch.empty(M, D, D, **f32))
        grad_rho_partial = torch.empty(M, _lib.NRHO, **f32)
        glam_partial = torch.empty(L, M, **f32)
        gt_partial = torch.empty(M, **f32)
        grad = torch.empty(G, _lib.NPARAM, **f32)
        wsp = lib.workspace(M, D, S)
        lib.glad_backward(Gout.contiguous(), S, params, init_diag, L, Z, half, U, beta, lam, lam_in, bufs[0], bufs[1],
                          grad_rho_partial, glam_partial, gt_partial, grad, wsp, mode, groups=G)
        return None, grad, None, None, None, None


def glad_grouped(Sb: Tensor, params: Tensor, lambda_init: float = 1, L: int = 15, INIT_DIAG: int = 0,
                 sqrt_mode: Optional[str] = None) -> Tensor:
    """`glad` for G independent problems at once: Sb (M, D, D) with M a multiple of G = params.shape[0]; params (G, 42) in the
    packed layout of GladParams.packed() (include/uglad_hip.h).  Group g = matrices [g M/G, (g+1) M/G): its own parameters,
    its own batch norm and lambda sequence -- exactly what G separate `glad` calls would compute."""
    if sqrt_mode is None:
        sqrt_mode = DEFAULT_SQRT_MODE
    if sqrt_mode not in _lib.SQRT_MODES:
        raise ValueError(f"sqrt_mode must be one of {sorted(_lib.SQRT_MODES)}")
    if INIT_DIAG not in (0, 1):
        raise ValueError("INIT_DIAG must be 0 or 1")
    if params.dim() != 2 or params.shape[1] != _lib.NPARAM or Sb.dim() != 3 or Sb.shape[0] % params.shape[0] != 0:
        raise ValueError("params must be (G, 42) and Sb (M, D, D) with M a multiple of G")
    Sb = Sb.detach().to(device=params.device, dtype=torch.float32).contiguous()
    return _GladGrouped.apply(Sb, params, int(L), int(INIT_DIAG), float(lambda_init), _lib.SQRT_MODES[sqrt_mode])
