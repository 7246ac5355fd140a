"""ctypes binding of libuglad_hip.so (C ABI: include/uglad_hip.h).

There is no CPU fallback: `get_lib()` raises if the gfx950 shared object is missing or no GPU is visible, and every
wrapper insists on contiguous fp32 tensors on the GPU.  PyTorch is only the owner of device memory and of the stream.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# UGLAD_LIB: development / profiling builds of the same library (e.g. with phase stamps); the product loads the in-tree build
LIB_PATH = os.environ.get("UGLAD_LIB") or os.path.join(_HERE, "csrc", "libuglad_hip.so")

SQRT_MODES = {"exact": 0, "ns10": 1}
NPARAM = 42
NRHO = 28

_c_float_p = ctypes.c_void_p
_SIGS = {
    "uglad_version": ([], ctypes.c_int),
    "uglad_max_dim": ([], ctypes.c_int),
    "uglad_max_eig_dim": ([], ctypes.c_int),
    "uglad_set_matrix_iteration": ([ctypes.c_int], ctypes.c_int),
    "uglad_validated_cond": ([], ctypes.c_float),
    "uglad_cond_is_upper_bound": ([ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int], ctypes.c_int),
    "uglad_workspace_floats": ([ctypes.c_int, ctypes.c_int], ctypes.c_int),
    "uglad_init_theta": ([_c_float_p, _c_float_p, ctypes.c_int, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int),
    "uglad_init_theta_bwd": ([_c_float_p, _c_float_p, ctypes.c_int, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int),
    "uglad_lambda_init": ([_c_float_p, ctypes.c_float, _c_float_p, _c_float_p, ctypes.c_void_p], ctypes.c_int),
    "uglad_cell_fwd": ([_c_float_p] * 11 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int),
    "uglad_cell_fwd_stage2": ([_c_float_p] * 11 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int),
    "uglad_sum_partials": ([_c_float_p, ctypes.c_int, _c_float_p, ctypes.c_void_p], ctypes.c_int),
    "uglad_lambda_step": ([_c_float_p, ctypes.c_float, _c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_void_p], ctypes.c_int),
    "uglad_cell_bwd": ([_c_float_p] * 12 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int),
    "uglad_loss_fwd": ([_c_float_p, _c_float_p, ctypes.c_int, _c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int),
    "uglad_loss_bwd": ([_c_float_p, _c_float_p, _c_float_p, ctypes.c_int, _c_float_p, _c_float_p, ctypes.c_float, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int),
    "uglad_finish_grads": ([_c_float_p] * 6 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int),
    "uglad_glad_forward": ([_c_float_p, _c_float_p, ctypes.c_float, ctypes.c_int, ctypes.c_int, _c_float_p, ctypes.c_int] + [_c_float_p] * 9
                           + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int),
    "uglad_glad_backward": ([_c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int] + [_c_float_p] * 13
                            + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int),
    "uglad_glad_forward_grouped": ([_c_float_p, _c_float_p, ctypes.c_float, ctypes.c_int, ctypes.c_int, _c_float_p, ctypes.c_int]
                                   + [_c_float_p] * 9 + [ctypes.c_int] * 4 + [ctypes.c_void_p], ctypes.c_int),
    "uglad_glad_backward_grouped": ([_c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int] + [_c_float_p] * 13
                                    + [ctypes.c_int] * 4 + [ctypes.c_void_p], ctypes.c_int),
    "uglad_glad_forward_sharded": ([_c_float_p, _c_float_p, ctypes.c_float, ctypes.c_int, ctypes.c_int, _c_float_p, ctypes.c_int] + [_c_float_p] * 9
                                   + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p], ctypes.c_int),
    "uglad_rccl_unique_id": ([ctypes.c_void_p], ctypes.c_int),
    "uglad_rccl_comm_init": ([ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int),
    "uglad_rccl_comm_destroy": ([ctypes.c_void_p], ctypes.c_int),
    "uglad_rccl_comm_count": ([ctypes.c_void_p, ctypes.c_void_p], ctypes.c_int),
    "uglad_rccl_allreduce_sum": ([_c_float_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p], ctypes.c_int),
    "uglad_consensus_partial": ([_c_float_p, ctypes.c_int, ctypes.c_int, _c_float_p, _c_float_p, ctypes.c_void_p], ctypes.c_int),
    "uglad_consensus_combine": ([_c_float_p, _c_float_p, ctypes.c_int, _c_float_p, ctypes.c_void_p], ctypes.c_int),
    "uglad_symeig": ([_c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int),
    "uglad_covariance": ([_c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, _c_float_p, _c_float_p,
                          _c_float_p, ctypes.c_void_p], ctypes.c_int),
    "uglad_tridiagonalize": ([_c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int),
    "uglad_symeig_jacobi": ([_c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int),
    "uglad_conditional_mean": ([_c_float_p] * 9 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int),
    "uglad_partial_correlations": ([_c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int),
    "uglad_support_metrics": ([_c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p],
                              ctypes.c_int),
    "uglad_set_wide_mode": ([ctypes.c_int], ctypes.c_int),
}
EXPORTS = tuple(_SIGS)


class UgladError(RuntimeError):
    pass


class HipLib:
    """Thin typed wrapper over the C ABI.  `require_gpu=False` exists only so that tests can point the same wrapper at the
    host build of the kernels under tests/simt_emul (CPU tensors, no stream); the package itself never does that."""

    def __init__(self, path: str = LIB_PATH, require_gpu: bool = True):
        if not os.path.exists(path):
            raise UgladError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  uglad_amd has no CPU fallback."
            )
        self.path = path
        self.require_gpu = require_gpu
        self._dll = ctypes.CDLL(path)
        for name, (argtypes, restype) in _SIGS.items():
            fn = getattr(self._dll, name)
            fn.argtypes = argtypes
            fn.restype = restype
        self.max_dim = int(self._dll.uglad_max_dim())
        self.max_eig_dim = int(self._dll.uglad_max_eig_dim())
        self.validated_cond = float(self._dll.uglad_validated_cond())
        self.version = int(self._dll.uglad_version())

    # ------------------------------------------------------------------ helpers
    def _p(self, t: Optional[torch.Tensor]):
        if t is None:
            return None
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise UgladError("uglad_amd kernels take contiguous float32 tensors")
        if self.require_gpu and not t.is_cuda:
            raise UgladError("uglad_amd kernels take GPU tensors (no CPU fallback)")
        return ctypes.c_void_p(t.data_ptr())

    def _stream(self):
        if not self.require_gpu:
            return None
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _check(name: str, rc: int):
        if rc != 0:
            kind = {-1: "NULL pointer", -2: "unsupported dimension", -3: "unknown mode",
                    -4: "RCCL not loadable / an RCCL call failed"}.get(rc, f"hipError {rc}")
            raise UgladError(f"{name} failed: {kind}")

    def _call(self, name, *args):
        self._check(name, getattr(self._dll, name)(*args, self._stream()))

    def cond_is_upper_bound(self, M: int, D: int, training: bool, mode: int) -> bool:
        """Whether cond_max of a cell call of this shape is the Gershgorin upper bound (matrix-iteration path) or the condition number itself."""
        rc = int(self._dll.uglad_cond_is_upper_bound(int(M), int(D), int(bool(training)), int(mode)))
        if rc < 0:
            self._check("uglad_cond_is_upper_bound", rc)
        return rc == 1

    def set_wide_mode(self, mode: int) -> None:
        """-1 automatic, 0 never, 1 always (D > 128): many workgroups per matrix for few large matrices (include/uglad_hip.h)."""
        self._check("uglad_set_wide_mode", self._dll.uglad_set_wide_mode(int(mode)))

    def set_matrix_iteration(self, mode: int) -> None:
        """The cell as the reference's own Newton-Schulz matrix iteration on dense tile products (csrc/wide_ns.h): -1 (default) beyond
        max_eig_dim and for few matrices of 128 < D <= 256, 0 beyond max_eig_dim only, 1 for every D.  Size workspaces after setting it."""
        self._check("uglad_set_matrix_iteration", self._dll.uglad_set_matrix_iteration(int(mode)))

    def workspace(self, M: int, D: int, like: torch.Tensor) -> torch.Tensor:
        """Caller-owned scratch of uglad_workspace_floats(M, D) floats on `like`'s device."""
        n = int(self._dll.uglad_workspace_floats(int(M), int(D)))
        if n < 0:
            self._check("uglad_workspace_floats", n)
        return torch.empty(n, dtype=torch.float32, device=like.device)

    # ------------------------------------------------------------------ entry points
    def init_theta(self, S, params, init_diag, theta0, workspace):
        M, D, _ = S.shape
        self._call("uglad_init_theta", self._p(S), self._p(params), int(init_diag), self._p(theta0), self._p(workspace), M, D)

    def init_theta_bwd(self, theta0, G0, init_diag, gt_partial, workspace=None):
        M, D, _ = theta0.shape
        self._call("uglad_init_theta_bwd", self._p(theta0), self._p(G0), int(init_diag), self._p(gt_partial),
                   self._p(workspace), M, D)

    def lambda_init(self, params, lambda_init, lam_out, lam_in):
        self._call("uglad_lambda_init", self._p(params), float(lambda_init), self._p(lam_out), self._p(lam_in))

    def cell_fwd(self, S, Z_in, lam, params, Z_out, half_out, U_out, beta_out, normF_partial, workspace, mode, cond_max=None):
        """cond_max: (M,) running maximum of cond(b^T b + 4/lam I) per matrix, or None (include/uglad_hip.h)."""
        M, D, _ = S.shape
        self._call("uglad_cell_fwd", self._p(S), self._p(Z_in), self._p(lam), self._p(params), self._p(Z_out),
                   self._p(half_out), self._p(U_out), self._p(beta_out), self._p(normF_partial), self._p(cond_max),
                   self._p(workspace), M, D, int(mode))

    def cell_fwd_stage2(self, S, Z_in, lam, params, Z_out, half_out, U_out, beta_out, normF_partial, workspace, mode,
                        cond_max=None):
        M, D, _ = S.shape
        self._call("uglad_cell_fwd_stage2", self._p(S), self._p(Z_in), self._p(lam), self._p(params), self._p(Z_out),
                   self._p(half_out), self._p(U_out), self._p(beta_out), self._p(normF_partial), self._p(cond_max),
                   self._p(workspace), M, D, int(mode))

    def sum_partials(self, partials, out):
        self._call("uglad_sum_partials", self._p(partials), partials.numel(), self._p(out))

    def lambda_step(self, normF_sum, inv_M, lam_prev, params, lam_next, lam_in_next):
        self._call("uglad_lambda_step", self._p(normF_sum), float(inv_M), self._p(lam_prev), self._p(params),
                   self._p(lam_next), self._p(lam_in_next))

    def cell_bwd(self, G_next, S, Z_in, half, U, beta, lam, params, G_out, grad_rho_partial, glam_partial, mode,
                 workspace=None):
        M, D, _ = S.shape
        self._call("uglad_cell_bwd", self._p(G_next), self._p(S), self._p(Z_in), self._p(half), self._p(U), self._p(beta),
                   self._p(lam), self._p(params), self._p(G_out), self._p(grad_rho_partial), self._p(glam_partial),
                   self._p(workspace), M, D, int(mode))

    def loss_fwd(self, theta, S, struct, loss_partial, theta_inv, workspace):
        M, D, _ = theta.shape
        self._call("uglad_loss_fwd", self._p(theta), self._p(S), S.shape[0], self._p(struct), self._p(loss_partial),
                   self._p(theta_inv), self._p(workspace), M, D)

    def loss_bwd(self, theta, theta_inv, S, struct, g_up, scale, G_out):
        M, D, _ = theta.shape
        self._call("uglad_loss_bwd", self._p(theta), self._p(theta_inv), self._p(S), S.shape[0], self._p(struct),
                   self._p(g_up), float(scale), self._p(G_out), M, D)

    def finish_grads(self, gt_partial, grad_rho_partial, glam_partial, lam_in, params, grad, L, M):
        self._call("uglad_finish_grads", self._p(gt_partial), self._p(grad_rho_partial), self._p(glam_partial),
                   self._p(lam_in), self._p(params), self._p(grad), int(L), int(M))

    def glad_forward(self, S, params, lambda_init, init_diag, L, Z, half, U, beta, lam, lam_in, nf_partial, nf_sum, workspace,
                     mode, groups: int = 1, cond_max=None):
        M, D, _ = S.shape
        if groups == 1:
            self._call("uglad_glad_forward", self._p(S), self._p(params), float(lambda_init), int(init_diag), int(L), self._p(Z),
                       int(Z.shape[0]), self._p(half), self._p(U), self._p(beta), self._p(lam), self._p(lam_in),
                       self._p(nf_partial), self._p(nf_sum), self._p(cond_max), self._p(workspace), M, D, int(mode))
        else:
            self._call("uglad_glad_forward_grouped", self._p(S), self._p(params), float(lambda_init), int(init_diag), int(L),
                       self._p(Z), int(Z.shape[0]), self._p(half), self._p(U), self._p(beta), self._p(lam), self._p(lam_in),
                       self._p(nf_partial), self._p(nf_sum), self._p(cond_max), self._p(workspace), M, D, int(groups), int(mode))

    # the sharded pass in one call: `exchange` = (function pointer, context) with the uglad_allreduce_fn signature (include/uglad_hip.h)
    ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)

    def glad_forward_sharded(self, S, params, lambda_init, init_diag, L, Z, half, U, beta, lam, lam_in, nf_partial, nf_sum, workspace,
                             mode, m_global: int, exchange, cond_max=None):
        M, D, _ = S.shape
        fn, ctx = exchange
        self._call("uglad_glad_forward_sharded", self._p(S), self._p(params), float(lambda_init), int(init_diag), int(L), self._p(Z),
                   int(Z.shape[0]), self._p(half), self._p(U), self._p(beta), self._p(lam), self._p(lam_in), self._p(nf_partial),
                   self._p(nf_sum), self._p(cond_max), self._p(workspace), M, D, int(m_global), int(mode),
                   ctypes.cast(fn, ctypes.c_void_p), ctypes.c_void_p(ctx) if isinstance(ctx, int) else ctx)

    def rccl_unique_id(self) -> bytes:
        buf = ctypes.create_string_buffer(128)
        self._check("uglad_rccl_unique_id", self._dll.uglad_rccl_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
        return buf.raw

    def rccl_comm_init(self, unique_id: bytes, nranks: int, rank: int) -> int:
        """ncclCommInitRank on the current device; returns the communicator as an integer handle."""
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        comm = ctypes.c_void_p()
        self._check("uglad_rccl_comm_init", self._dll.uglad_rccl_comm_init(ctypes.cast(buf, ctypes.c_void_p), int(nranks), int(rank),
                                                                           ctypes.cast(ctypes.byref(comm), ctypes.c_void_p)))
        return int(comm.value)

    def rccl_comm_destroy(self, comm: int) -> None:
        self._check("uglad_rccl_comm_destroy", self._dll.uglad_rccl_comm_destroy(ctypes.c_void_p(comm)))

    def rccl_comm_count(self, comm: int) -> int:
        """ncclCommCount of a communicator made by rccl_comm_init."""
        n = ctypes.c_int(0)
        self._check("uglad_rccl_comm_count", self._dll.uglad_rccl_comm_count(ctypes.c_void_p(comm), ctypes.cast(ctypes.byref(n), ctypes.c_void_p)))
        return int(n.value)

    def rccl_exchange(self, comm: int):
        """(function pointer, context) for glad_forward_sharded: ncclAllReduce issued from the library on the compute stream."""
        return ctypes.cast(self._dll.uglad_rccl_allreduce_sum, ctypes.c_void_p), int(comm)

    def glad_backward(self, G_L, S, params, init_diag, L, Z, half, U, beta, lam, lam_in, gbuf0, gbuf1, grad_rho_partial,
                      glam_partial, gt_partial, grad, workspace, mode, groups: int = 1):
        M, D, _ = S.shape
        args = (self._p(G_L), self._p(S), self._p(params), int(init_diag), int(L), self._p(Z), self._p(half), self._p(U),
                self._p(beta), self._p(lam), self._p(lam_in), self._p(gbuf0), self._p(gbuf1), self._p(grad_rho_partial),
                self._p(glam_partial), self._p(gt_partial), self._p(grad), self._p(workspace), M, D)
        if groups == 1:
            self._call("uglad_glad_backward", *args, int(mode))
        else:
            self._call("uglad_glad_backward_grouped", *args, int(groups), int(mode))

    def consensus_partial(self, theta_K, absmin, signsum):
        K, D, _ = theta_K.shape
        self._call("uglad_consensus_partial", self._p(theta_K), K, D, self._p(absmin), self._p(signsum))

    def consensus_combine(self, absmin, signsum, out):
        D = absmin.shape[-1]
        self._call("uglad_consensus_combine", self._p(absmin), self._p(signsum), D, self._p(out))

    def tridiagonalize(self, A0, A1, lam, R, workspace):
        M, D, _ = A0.shape
        self._call("uglad_tridiagonalize", self._p(A0), self._p(A1), self._p(lam), self._p(R), self._p(workspace), M, D)

    def covariance(self, X, normalize: bool = False, eval_offset: float = 0.1, repair: bool = True,
                   return_min_eig: bool = False):
        """(K,N,D) tables on the device -> (K,D,D) covariances (uglad_covariance).  With `return_min_eig` also the smallest
        eigenvalue of every covariance BEFORE the repair, as the device's fp32 solver saw it ((K,) tensor)."""
        K, N, D = X.shape
        S = torch.empty(K, D, D, dtype=torch.float32, device=X.device)
        scratch = torch.empty(K * D * D + K * D, dtype=torch.float32, device=X.device) if repair else None
        wsp = self.workspace(K, D, X) if repair else None  # both must outlive the enqueue
        self._call("uglad_covariance", self._p(X), K, N, D, int(bool(normalize)), float(eval_offset), self._p(S), self._p(scratch),
                   self._p(wsp))
        if return_min_eig:
            if not repair:
                raise UgladError("return_min_eig needs repair=True (the eigenvalues come from the repair's solver run)")
            return S, scratch[K * D * D:].reshape(K, D)[:, 0].clone()
        return S

    def conditional_mean(self, precision, mean, observed, values, clip01: bool = False):
        """uglad_conditional_mean: (K,D,D), (K,D), (K,D) mask, (K,D) -> full_mean (K,D), cond_cov (K,D,D), log_pdf (K)."""
        K, D, _ = precision.shape
        f32 = dict(dtype=torch.float32, device=precision.device)
        full_mean, cond_cov, log_pdf = torch.empty(K, D, **f32), torch.empty(K, D, D, **f32), torch.empty(K, **f32)
        scratch = torch.empty(K, D, D, **f32)
        wsp = self.workspace(K, D, precision)
        self._call("uglad_conditional_mean", self._p(precision), self._p(mean), self._p(observed), self._p(values),
                   self._p(full_mean), self._p(cond_cov), self._p(log_pdf), self._p(scratch), self._p(wsp), K, D, int(bool(clip01)))
        return full_mean, cond_cov, log_pdf

    def partial_correlations(self, precision):
        K, D, _ = precision.shape
        rho = torch.empty_like(precision)
        self._call("uglad_partial_correlations", self._p(precision), self._p(rho), K, D)
        return rho

    def support_metrics(self, true_theta, pred_theta, beta: int = 1):
        """uglad_support_metrics: two (K,D,D) fp32 tensors -> (K, 11) float64 tensor on the device."""
        K, D, _ = pred_theta.shape
        out = torch.empty(K, 11, dtype=torch.float64, device=pred_theta.device)
        self._call("uglad_support_metrics", self._p(true_theta), self._p(pred_theta), ctypes.c_void_p(out.data_ptr()), K, D,
                   int(beta))
        return out

    def symeig(self, A, U, beta, jacobi: bool = False):
        M, D, _ = A.shape
        if jacobi:
            self._call("uglad_symeig_jacobi", self._p(A), self._p(U), self._p(beta), M, D)
        else:
            wsp = self.workspace(M, D, A)  # must outlive the enqueue (the caching allocator keeps it valid for the stream)
            self._call("uglad_symeig", self._p(A), self._p(U), self._p(beta), self._p(wsp), M, D)


_instance: Optional[HipLib] = None


def get_lib() -> HipLib:
    """The process-wide library handle.  Fails loudly when the HIP build or the GPU is missing."""
    global _instance
    if _instance is None:
        if not torch.cuda.is_available():
            raise UgladError("uglad_amd needs an MI355X visible to PyTorch-ROCm; there is no CPU fallback")
        _instance = HipLib(LIB_PATH, require_gpu=True)
    return _instance


def device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device())
