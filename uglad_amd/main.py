"""uGLAD's training/inference surface on MI355X.  Mirrors the hot-path half of `uglad/main.py`:

    uGLAD_GL / uGLAD_multitask        main.py:34-226    sklearn-GraphicalLassoCV-style wrappers (covariance_, precision_, location_)
    init_uGLAD / forward_uGLAD / loss_uGLAD   main.py:233-335
    run_uGLAD_direct / _CV / _missing / _multitask   main.py:338-789   epoch loops (host Python, as in the reference)
    mean_imputation / get_final_precision_from_batch main.py:647-716

Same names, argument meaning, defaults and return shapes.  Deliberate differences: invalid arguments raise ValueError
instead of sys.exit(0) (main.py:137,667,708); epochs < 10 works (the reference divides by zero, main.py:386); nothing is
plotted (main.py:416 writes ./loss_curve.png); tensors live on the GPU until `fit` stores numpy attributes.  Additive
only: `sqrt_mode=`, `predict()`, and sharding of the multi-task / missing-data batch over the ranks of an initialised
torch.distributed process group (one process per GPU, RCCL).
"""
from __future__ import annotations

import contextlib
import copy
from time import time
from typing import Optional

import numpy as np
import torch

from . import _lib
from .dist import Collective, get_collective
from .glad import glad
from .glad.glad_params import GladParams
from .utils import prepare_data
from .utils.metrics import report_metrics_all  # noqa: F401  (host-side API parity; the drivers count on the device)


# ============================================================================================ loss
class _GlassoLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, S, struct, divisor):
        lib = _lib.get_lib()
        M, D, _ = theta.shape
        f32 = dict(dtype=torch.float32, device=theta.device)
        theta = theta.contiguous()
        partial = torch.empty(M, **f32)
        theta_inv = torch.empty(M, D, D, **f32)
        wsp = lib.workspace(M, D, theta)
        lib.loss_fwd(theta, S, struct, partial, theta_inv, wsp)
        total = torch.empty(1, **f32)
        lib.sum_partials(partial, total)
        ctx.save_for_backward(theta, theta_inv, S, struct if struct is not None else torch.empty(0, **f32))
        ctx.has_struct = struct is not None
        ctx.scale = 1.0 / float(divisor)
        return (total * ctx.scale).reshape(())

    @staticmethod
    def backward(ctx, g):
        lib = _lib.get_lib()
        theta, theta_inv, S, struct = ctx.saved_tensors
        G = torch.empty_like(theta)
        g_up = g.detach().to(torch.float32).reshape(1).contiguous()
        lib.loss_bwd(theta, theta_inv, S, struct if ctx.has_struct else None, g_up, ctx.scale, G)
        return G, None, None, None


def loss_uGLAD(theta: torch.Tensor, S: torch.Tensor, struct_theta: Optional[torch.Tensor] = None,
               batch_divisor: Optional[int] = None) -> torch.Tensor:
    """Glasso objective sum_b(-logdet Theta_b + tr(S_b Theta_b)) / B with B = S.shape[0] (ref main.py:289-335), plus the
    log-cosh structure penalty when `struct_theta` is given.  S may be (1, D, D) against Theta (K, D, D) (the
    missing-data call, main.py:620-622; the divisor is then 1).  `batch_divisor` overrides B for a sharded batch."""
    dev = theta.device
    S = S.detach().to(device=dev, dtype=torch.float32).contiguous()
    if S.dim() == 2:
        S = S[None]
    if S.shape[0] not in (1, theta.shape[0]):
        raise ValueError("S must hold one matrix or one per precision matrix")
    if struct_theta is not None:
        struct_theta = struct_theta.detach().to(device=dev, dtype=torch.float32)
        if struct_theta.dim() == 2:
            struct_theta = struct_theta[None]
        if struct_theta.shape[0] == 1 and S.shape[0] > 1:  # one prior for the whole batch: the reference's (1 - struct) - eye broadcasts it (main.py:327-328)
            struct_theta = struct_theta.expand(S.shape[0], -1, -1)
        struct_theta = struct_theta.contiguous()
        if struct_theta.shape != S.shape:
            raise ValueError("struct_theta must have the shape of S (or hold one matrix for the whole batch)")
    B = S.shape[0] if batch_divisor is None else batch_divisor
    return _GlassoLoss.apply(theta, S, struct_theta, B)


def _loss_per_matrix(theta: torch.Tensor, S: torch.Tensor) -> torch.Tensor:
    """-logdet Theta_b + tr(S_b Theta_b) for every matrix of the batch, (M,) on the device, no autograd (the test-fold losses
    of the batched CV driver)."""
    lib = _lib.get_lib()
    M, D, _ = theta.shape
    theta = theta.detach().contiguous()
    S = S.detach().to(device=theta.device, dtype=torch.float32).contiguous()
    partial = torch.empty(M, dtype=torch.float32, device=theta.device)
    theta_inv = torch.empty_like(theta)
    wsp = lib.workspace(M, D, theta)
    lib.loss_fwd(theta, S, None, partial, theta_inv, wsp)
    return partial


# ============================================================================================ model / forward
def init_uGLAD(lr: float, theta_init_offset: float = 1.0, nF: int = 3, H: int = 3):
    """GladParams on the GPU + Adam (ref main.py:233-249)."""
    model = GladParams(theta_init_offset=theta_init_offset, nF=nF, H=H, device=_lib.device())
    optimizer = glad.get_optimizers(model, lr_glad=lr)
    return model, optimizer


def forward_uGLAD(Sb, model_glad, L: int = 15, INIT_DIAG: int = 0, loss_Sb=None, struct_theta=None,
                  sqrt_mode: Optional[str] = None, collective: Optional[Collective] = None,
                  global_batch: Optional[int] = None):
    """predTheta = glad(Sb), loss = glasso loss on Sb (or on loss_Sb) -- ref main.py:252-286."""
    dev = next(model_glad.parameters()).device
    Sb = Sb.to(dev)
    predTheta = glad.glad(Sb, model_glad, L=L, INIT_DIAG=INIT_DIAG, sqrt_mode=sqrt_mode, collective=collective,
                          global_batch=global_batch)
    if loss_Sb is None:
        # divisor = number of matrices in the WHOLE batch (main.py:306,315), also when this rank holds a shard of it
        loss = loss_uGLAD(predTheta, Sb, struct_theta=struct_theta, batch_divisor=global_batch)
    else:
        loss = loss_uGLAD(predTheta, loss_Sb, struct_theta=struct_theta)
    return predTheta, loss


def _to_dev(x):
    return prepare_data.convert_to_torch(x, req_grad=False, device=_lib.device())


_DEVICE_COVARIANCE = False  # set for the duration of a fit()/predict() by uGLAD_GL / uGLAD_multitask(device_covariance=True)


@contextlib.contextmanager
def device_covariance(enabled: bool = True):
    """Within this context the drivers form the covariances (and the reference's eigenvalue repair) on the GPU
    (`uglad_covariance`) instead of in fp64 numpy on the host (prepare_data.py:328-356)."""
    global _DEVICE_COVARIANCE
    saved, _DEVICE_COVARIANCE = _DEVICE_COVARIANCE, bool(enabled)
    try:
        yield
    finally:
        _DEVICE_COVARIANCE = saved


REPAIR_THRESHOLD = 1e-6  # prepare_data.py:347: "min eig <= 1e-6 -> S += (offset - min eig) I"
REPAIR_BAND = 1e-4       # fp32 eigenvalues within this (relative to ||S||_2-ish scale) of the threshold are re-decided in fp64


def _device_covariance_group(X: np.ndarray, eval_offset: float) -> torch.Tensor:
    """(K, N, D) equally-shaped tables -> (K, D, D) repaired covariances on the device.

    The reference decides its repair from fp64 eigenvalues (prepare_data.py:343-352); the device sees fp32 eigenvalues of an
    fp32 covariance, whose absolute error is ~1e-7 ||S||.  Where the device's smallest eigenvalue lies within REPAIR_BAND of
    the threshold the decision could flip -- an O(offset) discontinuity -- so exactly those matrices (rare: nearly singular
    with a minimum eigenvalue of ~1e-6) are re-decided from an fp64 eigvalsh on the host and corrected in place."""
    lib = _lib.get_lib()
    Xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).to(_lib.device())
    S, mn = lib.covariance(Xd, normalize=False, eval_offset=eval_offset, repair=True, return_min_eig=True)
    mn = mn.cpu().numpy().astype(np.float64)
    scale = np.maximum(1.0, S.diagonal(dim1=1, dim2=2).abs().sum(1).cpu().numpy())  # trace >= ||S||_2 for a PSD matrix
    for k in np.nonzero(np.abs(mn - REPAIR_THRESHOLD) <= REPAIR_BAND * scale)[0]:
        applied = mn[k] <= REPAIR_THRESHOLD
        Sk = S[k].double().cpu().numpy()
        if applied:
            Sk = Sk - (eval_offset - mn[k]) * np.eye(Sk.shape[0])  # undo the device's shift, decide again
        mn64 = float(np.linalg.eigvalsh(0.5 * (Sk + Sk.T)).min())
        if mn64 <= REPAIR_THRESHOLD:
            Sk = Sk + (eval_offset - mn64) * np.eye(Sk.shape[0])
        S[k] = torch.from_numpy(Sk.astype(np.float32)).to(S.device)
    return S


def _covariance(Xb, eval_offset):
    """Tables (K,N,D) -> (K,D,D) fp32 covariances on the device, repaired as prepare_data.get_covariance does.
    Under device_covariance the tables are grouped by shape (missing mode's row-subsampled folds and multitask tables may
    differ in length) and every group of equal shape is one launch of the device front-end."""
    if _DEVICE_COVARIANCE:
        tables = [np.asarray(x) for x in Xb]
        max_dim = _lib.get_lib().max_eig_dim  # (the device front-end's kernels: wider tables take the host path below)
        if tables and all(t.ndim == 2 and t.shape[1] <= max_dim and t.dtype != object for t in tables):
            groups = {}
            for i, t in enumerate(tables):
                groups.setdefault(t.shape, []).append(i)
            D = tables[0].shape[1]
            if all(shape[1] == D for shape in groups):
                out = torch.empty(len(tables), D, D, dtype=torch.float32, device=_lib.device())
                for idx in groups.values():
                    out[idx] = _device_covariance_group(np.stack([tables[i] for i in idx]), eval_offset)
                return out
    return _to_dev(prepare_data.get_covariance(Xb, offset=eval_offset))


def _print_every(EPOCHS: int) -> int:
    return max(1, int(EPOCHS / 10))


def _allreduce_grads(model, loss, coll: Collective):
    """Exchange (ii): SUM of the 42 gradients and of the loss over ranks, in one 43-float message."""
    if coll.world_size == 1:
        return loss.detach()
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()] + [loss.detach().reshape(1)])
    coll.all_reduce_sum(flat)
    off = 0
    for p in model.parameters():
        n = p.numel()
        p.grad.copy_(flat[off:off + n].reshape(p.shape))
        off += n
    return flat[off]


# ============================================================================================ drivers
def run_uGLAD_direct(Xb, trueTheta=None, eval_offset=0.1, EPOCHS=250, lr=0.002, INIT_DIAG=0, L=15, VERBOSE=True,
                     sqrt_mode=None):
    """Direct mode: one table, one covariance, EPOCHS Adam steps on the glasso loss (ref main.py:338-425).
    Passing trueTheta adds the reference's log-cosh structure penalty to the loss (main.py:398)."""
    Sb = _covariance(Xb, eval_offset)
    if trueTheta is not None:
        trueTheta = _to_dev(trueTheta)
    B = Sb.shape[0]
    model_glad, optimizer_glad = init_uGLAD(lr=lr, theta_init_offset=1.0, nF=3, H=3)
    PRINT_EVERY = _print_every(EPOCHS)
    predTheta = None
    # The reference stops at the first NaN loss BEFORE that epoch's backward/step (main.py:401-406).  Asking the device for
    # the loss every epoch would stall the host behind the GPU, so the NaN flag of an epoch travels to pinned host memory
    # asynchronously and is looked at one epoch later; the speculative epoch is then undone (the 42 parameters are snapshotted
    # before every step), which leaves exactly the model the reference returns.
    lagged = Sb.is_cuda
    params = list(model_glad.parameters())
    pending = None  # (epoch, event, pinned flag, predTheta of that epoch, snapshot taken before that epoch's step)

    def snapshot():  # one launch: the 42 parameters packed (the optimiser is not returned, its state needs no rescue)
        return torch.cat([p.detach().reshape(-1) for p in params])

    def restore(snap):
        with torch.no_grad():
            off = 0
            for p in params:
                p.copy_(snap[off:off + p.numel()].reshape(p.shape))
                off += p.numel()

    def nan_epoch(p):
        p[1].synchronize()
        return bool(p[2].item())

    stopped = None
    for e in range(EPOCHS):
        optimizer_glad.zero_grad()
        predTheta_e, loss = forward_uGLAD(Sb, model_glad, L=L, INIT_DIAG=INIT_DIAG, struct_theta=trueTheta,
                                          sqrt_mode=sqrt_mode, collective=Collective())
        if lagged:
            if pending is not None and nan_epoch(pending):
                stopped = pending
                break
            flag = torch.empty(1, dtype=torch.bool).pin_memory()
            flag.copy_(torch.isnan(loss.detach()).reshape(1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            pending = (e, ev, flag, predTheta_e, snapshot())
        elif torch.isnan(loss):
            print(f"Warning: NaN loss encountered at epoch {e}. Try updating the parameters and train.")
            predTheta = predTheta_e
            break
        predTheta = predTheta_e
        loss.backward()
        if not e % PRINT_EVERY and VERBOSE:
            print(f"epoch:{e}/{EPOCHS} loss:{loss.item()}")
        optimizer_glad.step()
    if lagged and stopped is None and pending is not None and nan_epoch(pending):
        stopped = pending  # the very last epoch was the NaN one
    if stopped is not None:
        print(f"Warning: NaN loss encountered at epoch {stopped[0]}. Try updating the parameters and train.")
        restore(stopped[4])
        predTheta = stopped[3]
    compare_theta = None
    if trueTheta is not None and predTheta is not None:
        compare_theta = device_report_metrics(trueTheta, predTheta)[B - 1]  # (the reference keeps the last matrix's dict)
        if VERBOSE:
            print(f"Compare - {compare_theta}")
    return predTheta, compare_theta, model_glad


def _kfold_indices(n: int, k: int):
    """sklearn KFold(n_splits=k) without shuffling: (train, test) index pairs."""
    if k < 2 or k > n:
        raise ValueError(f"k_fold must be in [2, n_samples], got {k}")
    sizes = np.full(k, n // k, dtype=int)
    sizes[: n % k] += 1
    idx = np.arange(n)
    cur = 0
    for s in sizes:
        test = idx[cur:cur + s]
        yield np.concatenate([idx[:cur], idx[cur + s:]]), test
        cur += s


def _cv_fold(_k, Sb_train, Sb_test, model_glad, optimizer_glad, EPOCHS, INIT_DIAG, L, VERBOSE, sqrt_mode):
    """One fold of run_uGLAD_CV: EPOCHS x {training step on the train-fold covariance, no_grad forward on the test fold}.
    The best-so-far test loss and the parameters that go with it are tracked ON THE DEVICE (a select per parameter tensor
    instead of the reference's host-side comparison + deepcopy, main.py:506-518), so an epoch needs no device-to-host copy
    and the host can run ahead of the GPU; the selection is the reference's: strictly smaller test loss, snapshot taken
    AFTER this epoch's optimiser step, NaN never wins."""
    one = Collective()
    params = list(model_glad.parameters())
    best_loss = torch.full((), float("inf"), dtype=torch.float32, device=Sb_train.device)
    best_state = torch.cat([p.detach().reshape(-1) for p in params])  # the 42 parameters, packed
    PRINT_EVERY = _print_every(EPOCHS)
    for e in range(EPOCHS):
        optimizer_glad.zero_grad()
        _, loss_train = forward_uGLAD(Sb_train, model_glad, L=L, INIT_DIAG=INIT_DIAG, sqrt_mode=sqrt_mode, collective=one)
        with torch.no_grad():
            _, loss_test = forward_uGLAD(Sb_test, model_glad, L=L, INIT_DIAG=INIT_DIAG, sqrt_mode=sqrt_mode, collective=one)
        loss_train.backward()
        optimizer_glad.step()
        with torch.no_grad():
            lt = loss_test.reshape(())
            improved = lt < best_loss
            best_loss = torch.where(improved, lt, best_loss)
            best_state = torch.where(improved, torch.cat([p.reshape(-1) for p in params]), best_state)
        if not e % PRINT_EVERY and VERBOSE:
            print(f"Fold {_k}: epoch:{e}/{EPOCHS} test-loss:{float(loss_test.item())}")
    best = float(best_loss.item())
    best_model = None
    if best < np.inf:
        best_model = copy.deepcopy(model_glad)
        with torch.no_grad():
            off = 0
            for p in best_model.parameters():
                p.copy_(best_state[off:off + p.numel()].reshape(p.shape))
                off += p.numel()
    return {"test_loss": best, "model": best_model}


def _cv_folds_batched(folds, EPOCHS, lr, INIT_DIAG, L, VERBOSE, sqrt_mode):
    """All folds of CV mode as ONE batch: fold k = group k of a grouped pass, with its own 42 parameters and
    its own lambda sequence (glad_grouped); one Adam over the (k, 42) parameter table (Adam is entrywise, so every fold gets
    exactly the updates of its own optimiser); the best test loss per fold and the parameters that go with it are selected on
    the device.  Per epoch: one grouped training pass, one grouped no_grad pass on the test folds, no host synchronisation."""
    from .glad.glad import glad_grouped

    k = len(folds)
    S_train = torch.cat([f[1] for f in folds])  # (k, D, D): one train-fold covariance per group
    S_test = torch.cat([f[2] for f in folds])
    S_both = torch.cat([S_train, S_test])
    P = torch.nn.Parameter(torch.stack([f[3].packed().detach() for f in folds]))  # models were initialised in fold order
    opt = glad.get_optimizers(_ParamTable(P), lr_glad=lr)
    best_loss = torch.full((k,), float("inf"), dtype=torch.float32, device=P.device)
    best_P = P.detach().clone()
    PRINT_EVERY = _print_every(EPOCHS)
    for e in range(EPOCHS):
        opt.zero_grad()
        # train folds and test folds ride in ONE grouped pass of 2k groups (a small pass is bound by the latency of its
        # kernels, not by how many matrices it carries); the test groups see the same parameters, detached
        theta = glad_grouped(S_both, torch.cat([P, P.detach()]), L=L, INIT_DIAG=INIT_DIAG, sqrt_mode=sqrt_mode)
        loss_train = loss_uGLAD(theta[:k], S_train, batch_divisor=1)  # sum of the folds' losses: d/dP[k] is fold k's own gradient
        with torch.no_grad():
            loss_test = _loss_per_matrix(theta[k:], S_test)
        loss_train.backward()
        opt.step()
        with torch.no_grad():
            improved = loss_test < best_loss  # (NaN never wins; snapshot AFTER this epoch's step, as in main.py:506,518)
            best_loss = torch.where(improved, loss_test, best_loss)
            best_P = torch.where(improved[:, None], P.detach(), best_P)
        if not e % PRINT_EVERY and VERBOSE:
            print(f"epoch:{e}/{EPOCHS} test-loss per fold: {[float(v) for v in loss_test.cpu()]}")
    results = {}
    bl = best_loss.cpu().numpy()
    for i, f in enumerate(folds):
        model = None
        if bl[i] < np.inf:
            model = copy.deepcopy(f[3])
            with torch.no_grad():
                off = 0
                for p in model.parameters():
                    p.copy_(best_P[i, off:off + p.numel()].reshape(p.shape))
                    off += p.numel()
        results[f[0]] = {"test_loss": float(bl[i]), "model": model}
    return results


class _ParamTable(torch.nn.Module):
    """A module around one (G, 42) parameter table, so that get_optimizers() can be handed the usual `.parameters()`."""

    def __init__(self, table: torch.nn.Parameter):
        super().__init__()
        self.table = table


def run_uGLAD_CV(Xb, trueTheta=None, eval_offset=0.1, EPOCHS=250, lr=0.002, INIT_DIAG=0, L=15, VERBOSE=True, k_fold=5,
                 sqrt_mode=None, parallel_folds: bool = False, batched_folds: bool = False):
    """k-fold CV mode (ref main.py:428-550): per fold a fresh model, per epoch one training step on the train-fold
    covariance and one no_grad forward on the test fold; the model with the best test loss over all folds is run on the
    full covariance.

    batched_folds (additive): all folds in ONE grouped batch -- per-fold parameters and lambda sequences inside
    the kernels (`_cv_folds_batched`), no host synchronisation per epoch.

    parallel_folds (additive): the folds are independent problems of one matrix each -- a single small matrix
    keeps one CU busy for a few hundred microseconds per kernel -- so each fold trains in its own host thread on its own HIP
    stream and the GPU runs them side by side.  The arithmetic of a fold is untouched (same kernels, same order), so the
    result is bit-identical to the sequential run; the models are still initialised in fold order, which keeps the draws
    from torch's global RNG those of the reference."""
    Sb = _covariance(Xb, eval_offset)
    if trueTheta is not None:
        trueTheta = _to_dev(trueTheta)
    one = Collective()
    B = 1
    folds = []
    for _k, (train, test) in enumerate(_kfold_indices(Xb[0].shape[0], k_fold)):
        Sb_train = _covariance(Xb[0][train][None], eval_offset)
        Sb_test = _covariance(Xb[0][test][None], eval_offset)
        model_glad, optimizer_glad = init_uGLAD(lr=lr, theta_init_offset=1.0, nF=3, H=3)
        folds.append((_k, Sb_train, Sb_test, model_glad, optimizer_glad))
    results = {}
    if batched_folds and len(folds) > 1:
        results = _cv_folds_batched(folds, EPOCHS, lr, INIT_DIAG, L, VERBOSE, sqrt_mode)
    elif parallel_folds and Sb.is_cuda and len(folds) > 1:
        import threading

        main_stream = torch.cuda.current_stream()
        streams = [torch.cuda.Stream(device=Sb.device) for _ in folds]
        errors = []

        def work(fold, stream):
            try:
                torch.cuda.set_device(Sb.device)
                with torch.cuda.stream(stream):
                    results[fold[0]] = _cv_fold(*fold, EPOCHS, INIT_DIAG, L, VERBOSE, sqrt_mode)
            except BaseException as exc:  # surfaced in the caller's thread below
                errors.append(exc)

        for s in streams:
            s.wait_stream(main_stream)  # the covariances and the initial parameters were produced on the caller's stream
        threads = [threading.Thread(target=work, args=(f, s)) for f, s in zip(folds, streams)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for s in streams:
            main_stream.wait_stream(s)
        if errors:
            raise errors[0]
    else:
        for fold in folds:
            if VERBOSE:
                print(f"Fold num {fold[0]}")
            results[fold[0]] = _cv_fold(*fold, EPOCHS, INIT_DIAG, L, VERBOSE, sqrt_mode)
    results = {k: results[k] for k in sorted(results)}
    best_loss = np.inf
    model_glad = None
    for _k in results:
        if results[_k]["test_loss"] < best_loss:
            model_glad = results[_k]["model"]
            best_loss = results[_k]["test_loss"]
    if model_glad is None:
        raise ValueError("cross-validation produced no finite test loss")
    with torch.no_grad():
        predTheta, _ = forward_uGLAD(Sb, model_glad, L=L, INIT_DIAG=INIT_DIAG, sqrt_mode=sqrt_mode, collective=one)
    compare_theta = None
    if trueTheta is not None:
        compare_theta = device_report_metrics(trueTheta, predTheta)[B - 1]
        if VERBOSE:
            print(f"Comparison - {compare_theta}")
    return predTheta, compare_theta, model_glad


def mean_imputation(Xb: np.ndarray) -> np.ndarray:
    """NaN -> column mean; an all-NaN column is an error (ref main.py:647-670; ValueError instead of sys.exit)."""
    X = np.array(Xb[0], dtype=np.float64)
    with np.errstate(all="ignore"):
        col_mean = np.nanmean(X, axis=0)
    if np.isnan(col_mean).any():
        raise ValueError("One or more columns have all NaNs")
    r, c = np.where(np.isnan(X))
    X[r, c] = col_mean[c]
    return X[None]


def get_final_precision_from_batch(predTheta: torch.Tensor, type: str = "min",
                                   collective: Optional[Collective] = None) -> torch.Tensor:
    """Consensus over K precision matrices: sign by majority (ties -> +), magnitude = min_k |Theta_k| (ref main.py:673-716).
    With a sharded batch the two partial results are all-reduced (MIN, SUM) before being combined.
    type="mean" -- buggy in the reference (it broadcasts row 0 of the mean, main.py:705) -- is implemented as the
    entrywise mean of |Theta_k| and is a labelled deviation."""
    lib = _lib.get_lib()
    coll = collective if collective is not None else Collective()
    predTheta = predTheta.detach().contiguous()
    K, D, _ = predTheta.shape
    f32 = dict(dtype=torch.float32, device=predTheta.device)
    absmin = torch.empty(D, D, **f32)
    signsum = torch.empty(D, D, **f32)
    lib.consensus_partial(predTheta, absmin, signsum)
    coll.all_reduce_min(absmin)
    coll.all_reduce_sum(signsum)
    if type == "min":
        value = absmin
    elif type == "mean":
        tot = predTheta.abs().sum(0)
        cnt = torch.tensor([float(K)], **f32)
        coll.all_reduce_sum(tot)
        coll.all_reduce_sum(cnt)
        value = (tot / cnt).contiguous()
    else:
        raise ValueError(f"Enter valid type min/mean, currently {type}")
    out = torch.empty(D, D, **f32)
    lib.consensus_combine(value, signsum, out)
    return out.reshape(1, D, D)


def run_uGLAD_missing(Xb, trueTheta=None, eval_offset=0.1, EPOCHS=250, lr=0.002, INIT_DIAG=0, L=15, VERBOSE=True,
                      K_batch=3, sqrt_mode=None):
    """Missing-data mode (ref main.py:553-644): mean imputation, K row-subsampled covariances as one multi-task batch whose
    loss is taken against the single full-data covariance (divisor 1), consensus at the end.  Under torch.distributed the K
    sub-batches are sharded over the ranks (config 5: one per GPU)."""
    if K_batch == 0:
        K_batch = 3
    coll = get_collective()
    _check_shardable(K_batch, coll, "K_batch sub-sample covariances")
    Xb = mean_imputation(Xb)
    Sb = _covariance(Xb, eval_offset)
    folds = [tr for tr, _ in _kfold_indices(Xb[0].shape[0], K_batch)]
    lo, hi = coll.shard(K_batch)
    X_K = [Xb[0][idx] for idx in folds[lo:hi]]
    S_K = _covariance(X_K, eval_offset)
    if trueTheta is not None:
        trueTheta = _to_dev(trueTheta)
    model_glad, optimizer_glad = init_uGLAD(lr=lr, theta_init_offset=1.0, nF=3, H=3)
    _broadcast_model(model_glad, coll)
    PRINT_EVERY = _print_every(EPOCHS)
    predTheta = None
    for e in range(EPOCHS):
        optimizer_glad.zero_grad()
        predTheta, loss = forward_uGLAD(S_K, model_glad, L=L, INIT_DIAG=INIT_DIAG, loss_Sb=Sb, sqrt_mode=sqrt_mode,
                                        collective=coll, global_batch=K_batch)
        loss.backward()
        total = _allreduce_grads(model_glad, loss, coll)
        optimizer_glad.step()
        if not e % PRINT_EVERY and VERBOSE:
            print(f"epoch:{e}/{EPOCHS} loss:{float(total)}")
    predTheta = get_final_precision_from_batch(predTheta, type="min", collective=coll)
    compare_theta = None
    if trueTheta is not None:
        compare_theta = device_report_metrics(trueTheta[:1], predTheta[:1])[0]
        if VERBOSE:
            print(f"Comparison - {compare_theta}")
    return predTheta, compare_theta, model_glad


def _check_shardable(K: int, coll: Collective, what: str) -> None:
    """Every rank must own at least one matrix: an empty shard would fail locally while the other ranks wait in the first
    collective.  K and world_size are known everywhere, so every rank raises the same error before any exchange."""
    if K < coll.world_size:
        raise ValueError(f"{K} {what} cannot be sharded over {coll.world_size} ranks (need at least one per rank)")


def _broadcast_model(model, coll: Collective):
    """Replicate rank 0's freshly initialised 42 parameters (each rank draws its own otherwise)."""
    if coll.world_size == 1:
        return
    with torch.no_grad():
        flat = torch.cat([p.reshape(-1) for p in model.parameters()])
        if coll.rank != 0:
            flat.zero_()
        coll.all_reduce_sum(flat)
        off = 0
        for p in model.parameters():
            n = p.numel()
            p.copy_(flat[off:off + n].reshape(p.shape))
            off += n


def run_uGLAD_multitask(Xb, trueTheta=None, eval_offset=0.1, EPOCHS=250, lr=0.002, INIT_DIAG=0, L=15, VERBOSE=True,
                        sqrt_mode=None):
    """Multi-task mode (ref main.py:719-789): K tables -> K covariances -> ONE shared 42-parameter model trained on the
    batch.  Under torch.distributed every rank passes the full list and works on its contiguous slice; the per-step
    norm, the gradients and the final precision matrices are exchanged over RCCL, so every rank returns all K."""
    K = len(Xb)
    coll = get_collective()
    _check_shardable(K, coll, "tasks")
    lo, hi = coll.shard(K)
    Sb = _covariance(Xb[lo:hi], eval_offset)
    model_glad, optimizer_glad = init_uGLAD(lr=lr, theta_init_offset=1.0, nF=3, H=3)
    _broadcast_model(model_glad, coll)
    PRINT_EVERY = _print_every(EPOCHS)
    predTheta = None
    for e in range(EPOCHS):
        optimizer_glad.zero_grad()
        predTheta, loss = forward_uGLAD(Sb, model_glad, L=L, INIT_DIAG=INIT_DIAG, sqrt_mode=sqrt_mode, collective=coll,
                                        global_batch=K)
        loss.backward()
        total = _allreduce_grads(model_glad, loss, coll)
        optimizer_glad.step()
        if not e % PRINT_EVERY and VERBOSE:
            print(f"epoch:{e}/{EPOCHS} loss:{float(total)}")
    predTheta = coll.all_gather_cat(predTheta.detach())
    compare_theta = []
    if trueTheta is not None:
        compare_theta = device_report_metrics(np.asarray(trueTheta), predTheta)  # all K graphs in one launch
        if VERBOSE:
            for b, rM in enumerate(compare_theta):
                print(f"Metrics for graph {b}: {rM}\n")
    return predTheta, compare_theta, model_glad


# ============================================================================================ public classes
class uGLAD_GL(object):
    """Drop-in for the reference's `uGLAD_GL` (main.py:34-151): `fit` sets covariance_ (float64), precision_ (float32),
    location_, node_names_, model_glad and returns the metrics dict (or None) -- not self, like the reference."""

    def __init__(self, device_covariance: bool = False):
        """device_covariance (additive): form the input covariances + eigenvalue repair on the GPU."""
        super().__init__()
        self.device_covariance = bool(device_covariance)
        self.covariance_: Optional[np.ndarray] = None
        self.precision_: Optional[np.ndarray] = None
        self.location_: Optional[np.ndarray] = None
        self.model_glad: Optional[GladParams] = None
        self._fit_cfg = None

    def fit(self, X, true_theta=None, eval_offset=0.1, centered=False, epochs=250, lr=0.002, INIT_DIAG=0, L=15,
            verbose=True, k_fold=3, mode="direct", node_names=None, sqrt_mode=None, parallel_folds=False,
            batched_folds=False):
        start = time()
        if verbose:
            print("Running uGLAD")
        X = np.array(prepare_data.process_table(X, NORM="min_max", VERBOSE=verbose))
        M, D = X.shape
        Xb = X.reshape(1, M, D)
        true_theta_b = None if true_theta is None else np.asarray(true_theta).reshape(1, D, D)
        kw = dict(trueTheta=true_theta_b, eval_offset=eval_offset, EPOCHS=epochs, lr=lr, INIT_DIAG=INIT_DIAG, L=L,
                  VERBOSE=verbose, sqrt_mode=sqrt_mode)
        with device_covariance(self.device_covariance), glad.regime_monitor() as regime:
            if mode == "missing":
                pred_theta, compare_theta, model_glad = run_uGLAD_missing(Xb, K_batch=k_fold, **kw)
            elif mode == "cv" and k_fold >= 0:
                pred_theta, compare_theta, model_glad = run_uGLAD_CV(Xb, k_fold=k_fold, parallel_folds=parallel_folds,
                                                                         batched_folds=batched_folds, **kw)
            elif mode == "direct":
                pred_theta, compare_theta, model_glad = run_uGLAD_direct(Xb, **kw)
            else:
                raise ValueError(f"Please enter K-fold value in valid range [0, ), currently entered {k_fold}; check mode {mode}")
        self.covariance_ = prepare_data.empirical_covariance(X, assume_centered=centered)
        self.location_ = X.mean(axis=0)
        self.node_names_ = list(node_names) if node_names is not None else [f"node_{i}" for i in range(D)]
        if pred_theta is not None:
            self.precision_ = pred_theta[0].detach().cpu().numpy()
        if model_glad is not None:
            self.model_glad = model_glad
        self._fit_cfg = dict(L=L, INIT_DIAG=INIT_DIAG, eval_offset=eval_offset, sqrt_mode=sqrt_mode)
        # regime diagnostic (additive): the largest cond(b^T b + 4/lambda I) any pass of this fit saw; warns beyond the validated bound
        self.cond_max_ = regime.warn_if_outside("uGLAD_GL.fit", get_collective() if mode == "missing" else None)  # (the sharded mode)
        if verbose:
            print(f"Total runtime: {time() - start} secs\n")
        return compare_theta

    def predict(self, X=None, S=None) -> np.ndarray:
        """Inference-only pass: the trained 42 parameters applied to new data (no_grad forward, the reference's
        main.py:539-540 pattern).  Give a samples table X (processed like `fit` does) or covariance(s) S.
        Warns (UgladRegimeWarning) when the pass leaves the validated regime -- e.g. an un-normalised covariance given as S."""
        if self.model_glad is None:
            raise ValueError("call fit() first")
        cfg = self._fit_cfg
        if S is None:
            X = np.array(prepare_data.process_table(X, NORM="min_max", VERBOSE=False))
            S = prepare_data.get_covariance(X[None], offset=cfg["eval_offset"])
        S = np.asarray(S, dtype=np.float64)
        if S.ndim == 2:
            S = S[None]
        with torch.no_grad(), glad.regime_monitor() as regime:
            theta = glad.glad(_to_dev(S), self.model_glad, L=cfg["L"], INIT_DIAG=cfg["INIT_DIAG"],
                              sqrt_mode=cfg["sqrt_mode"], collective=Collective())
        self.predict_cond_max_ = regime.warn_if_outside("uGLAD_GL.predict")
        out = theta.cpu().numpy()
        return out[0] if out.shape[0] == 1 else out


class uGLAD_multitask(object):
    """Drop-in for the reference's `uGLAD_multitask` (main.py:155-226): K tables, one shared model, batched attributes."""

    def __init__(self, device_covariance: bool = False):
        super().__init__()
        self.device_covariance = bool(device_covariance)
        self.covariance_ = []
        self.precision_: Optional[np.ndarray] = None
        self.model_glad: Optional[GladParams] = None
        self._fit_cfg = None

    def fit(self, Xb, true_theta_b=None, eval_offset=0.1, centered=False, epochs=250, lr=0.002, INIT_DIAG=0, L=15,
            verbose=True, sqrt_mode=None):
        start = time()
        if verbose:
            print("Running uGLAD in multi-task mode")
        Xb = [np.array(prepare_data.process_table(X, NORM="min_max", VERBOSE=verbose)) for X in Xb]
        with device_covariance(self.device_covariance), glad.regime_monitor() as regime:
            pred_theta, compare_theta, model_glad = run_uGLAD_multitask(
                Xb, trueTheta=true_theta_b, eval_offset=eval_offset, EPOCHS=epochs, lr=lr, INIT_DIAG=INIT_DIAG, L=L,
                VERBOSE=verbose, sqrt_mode=sqrt_mode)
        self.covariance_ = np.array([prepare_data.empirical_covariance(X, assume_centered=centered) for X in Xb])
        self.precision_ = pred_theta.detach().cpu().numpy()
        self.model_glad = model_glad
        self._fit_cfg = dict(L=L, INIT_DIAG=INIT_DIAG, eval_offset=eval_offset, sqrt_mode=sqrt_mode)
        self.cond_max_ = regime.warn_if_outside("uGLAD_multitask.fit", get_collective())
        if verbose:
            print(f"Total runtime: {time() - start} secs\n")
        return compare_theta

    def predict(self, Xb=None, S=None) -> np.ndarray:
        """no_grad forward of the trained model on new tables (list) or covariances (K, D, D)."""
        if self.model_glad is None:
            raise ValueError("call fit() first")
        cfg = self._fit_cfg
        if S is None:
            Xb = [np.array(prepare_data.process_table(X, NORM="min_max", VERBOSE=False)) for X in Xb]
            S = prepare_data.get_covariance(Xb, offset=cfg["eval_offset"])
        with torch.no_grad(), glad.regime_monitor() as regime:
            theta = glad.glad(_to_dev(np.asarray(S, dtype=np.float64)), self.model_glad, L=cfg["L"],
                              INIT_DIAG=cfg["INIT_DIAG"], sqrt_mode=cfg["sqrt_mode"], collective=Collective())
        self.predict_cond_max_ = regime.warn_if_outside("uGLAD_multitask.predict")
        return theta.cpu().numpy()


# ============================================================================================ after the path
METRIC_KEYS = ("FDR", "TPR", "FPR", "SHD", "nnzTrue", "nnzPred", "precision", "recall", "Fbeta", "aupr", "auc")


def device_report_metrics(true_theta, pred_theta, beta: int = 1):
    """`report_metrics_all` (ref utils/metrics.py:25-108) for K (true, predicted) precision matrices at once, counted on the
    device (uglad_support_metrics): list of K dicts with the reference's keys, rounded to 3 decimals as the reference does."""
    lib = _lib.get_lib()
    dev = _lib.device()
    T = torch.as_tensor(np.asarray(true_theta.detach().cpu() if torch.is_tensor(true_theta) else true_theta).real,
                        dtype=torch.float32) if not (torch.is_tensor(true_theta) and true_theta.device == dev) else true_theta
    G = pred_theta if torch.is_tensor(pred_theta) else torch.as_tensor(np.asarray(pred_theta).real, dtype=torch.float32)
    T = T.detach().to(device=dev, dtype=torch.float32)
    G = G.detach().to(device=dev, dtype=torch.float32)
    if T.dim() == 2:
        T, G = T[None], G[None]
    if T.shape[-1] > lib.max_eig_dim:
        # beyond the counting kernel's size (one workgroup sweeps all D (D - 1) / 2 scores per true edge): the report is evaluated where the
        # reference evaluates it -- on the host, from the same definitions (utils/metrics.py = ref utils/metrics.py:25-108).  After the path, once
        # per fit; round 3 raised UgladError here AFTER all epochs of a fit(X, true_theta=...) at D > 256 had run.
        Tn, Gn = T.cpu().numpy(), G.cpu().numpy()
        return [report_metrics_all(Tn[k], Gn[k], beta=beta) for k in range(Tn.shape[0])]
    out = lib.support_metrics(T.contiguous(), G.contiguous(), beta=beta).cpu().numpy()
    return [{k: round(float(v), 3) for k, v in zip(METRIC_KEYS, row)} for row in out]


def get_partial_correlations(precision):
    """rho_ij = -p_ij / sqrt(p_ii p_jj), ones on the diagonal (ref main.py:796-821: its double loop fills the upper triangle
    with that formula and mirrors it).  (D,D) or (K,D,D).
    A host array (what the reference takes: `precision_` is numpy) is evaluated on the host in float64 like the reference -- an O(D^2)
    formula needs no GPU and loses nothing to fp32; a tensor already on the device goes through uglad_partial_correlations (fp32, K
    matrices per launch) and comes back as a device tensor."""
    if torch.is_tensor(precision) and precision.is_cuda:
        P = precision.detach().to(dtype=torch.float32)
        single = P.dim() == 2
        rho = _lib.get_lib().partial_correlations((P[None] if single else P).contiguous())
        return rho[0] if single else rho
    P = np.asarray(precision.detach().cpu() if torch.is_tensor(precision) else precision, dtype=np.float64)
    d = np.sqrt(np.diagonal(P, axis1=-2, axis2=-1))
    up = np.triu(-P / (d[..., :, None] * d[..., None, :]), k=1)  # the reference reads the upper triangle and mirrors it
    rho = up + np.swapaxes(up, -1, -2)
    idx = np.arange(P.shape[-1])
    rho[..., idx, idx] = 1.0
    return rho


def conditional_gaussian_batch(precision, mean, observed_mask, observed_values, clip01: bool = False):
    """K conditional-Gaussian problems on the device (uglad_conditional_mean): precision (K,D,D), mean (K,D), observed_mask (K,D)
    bool, observed_values (K,D) (read where observed) -> full_mean (K,D), cond_cov (K,D,D) (L_uu^-1 on the unobserved block,
    identity elsewhere), log_pdf (K) -- torch tensors on the device."""
    dev = _lib.device()
    f = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32).to(dev).contiguous() if not torch.is_tensor(a) \
        else a.detach().to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
    if np.shape(precision)[-1] > _lib.get_lib().max_eig_dim:
        return _conditional_gaussian_host(precision, mean, observed_mask, observed_values, clip01, dev)
    return _lib.get_lib().conditional_mean(f(precision), f(mean), f(observed_mask), f(observed_values), clip01=clip01)


def _conditional_gaussian_host(precision, mean, observed_mask, observed_values, clip01, dev):
    """D beyond the device solver of uglad_conditional_mean (the path's eigensolver, D <= 256): the reference's own formulation on the host in
    float64 (ref main.py:1176-1227: mean_u - L_uu^-1 L_uo (x_o - mean_o), conditional covariance L_uu^-1, the density at the MAP point), returned
    in the layout of the device entry point (full mean; L_uu^-1 on the unobserved block, identity elsewhere; log density)."""
    to64 = lambda a: (a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)).astype(np.float64)  # noqa: E731
    P, mu, mask, vals = to64(precision), to64(mean), to64(observed_mask) != 0, to64(observed_values)
    K, D = mu.shape
    full, cov, logp = np.empty((K, D)), np.empty((K, D, D)), np.empty(K)
    for k in range(K):
        o, u = np.where(mask[k])[0], np.where(~mask[k])[0]
        Luu_inv = np.linalg.inv(P[k][np.ix_(u, u)]) if u.size else np.zeros((0, 0))
        m = mu[k].copy()
        m[o] = vals[k][o]
        if u.size:
            m[u] = mu[k][u] - Luu_inv @ (P[k][np.ix_(u, o)] @ (vals[k][o] - mu[k][o]))
        if clip01:
            m = np.clip(m, 0.0, 1.0)
        full[k] = m
        cov[k] = np.eye(D)
        cov[k][np.ix_(u, u)] = Luu_inv
        # density of the conditional Gaussian at its own mean: (2 pi)^(-n_u / 2) det(L_uu)^(1/2)
        sign, ld = np.linalg.slogdet(P[k][np.ix_(u, u)]) if u.size else (1.0, 0.0)
        logp[k] = -0.5 * u.size * np.log(2.0 * np.pi) + 0.5 * ld if sign > 0 else np.nan
    t = lambda a: torch.as_tensor(a, dtype=torch.float32).to(dev)  # noqa: E731
    return t(full), t(cov), t(logp)


def conditional_gaussian_with_probabilities(precision, mean, observed_idx, observed_values):
    """Conditional mean, conditional covariance and density at the MAP point of a Gaussian given observed coordinates
    (ref main.py:1176-1227, same arguments and return values): (full_mean (n,), conditional_cov (n_u, n_u), pdf).
    The reference solves with scipy on the host; here the path's eigensolver does it on the device (fp32)."""
    precision = np.asarray(precision)
    mean = np.asarray(mean, dtype=np.float64)
    n = len(mean)
    observed_idx = [int(i) for i in observed_idx]
    mask = np.zeros(n, dtype=np.float32)
    mask[observed_idx] = 1.0
    vals = np.zeros(n, dtype=np.float64)
    vals[observed_idx] = np.asarray(observed_values, dtype=np.float64)
    full, cov, logp = conditional_gaussian_batch(precision[None], mean[None], mask[None], vals[None])
    unobs = [i for i in range(n) if mask[i] == 0.0]
    cov = cov[0].cpu().numpy().astype(np.float64)
    return full[0].cpu().numpy().astype(np.float64), cov[np.ix_(unobs, unobs)], float(np.exp(np.float64(logp[0].item())))


def compute_map_estimate(observed_nodes: dict, model_uGLAD) -> np.ndarray:
    """MAP estimate of all nodes given some observed ones, clamped to [0, 1] (ref main.py:1229-1260): uses the estimator's
    precision_, location_ and node_names_."""
    names = list(model_uGLAD.node_names_)
    idx = [names.index(k) for k in observed_nodes.keys()]
    n = len(names)
    mask = np.zeros(n, dtype=np.float32)
    mask[idx] = 1.0
    vals = np.zeros(n, dtype=np.float64)
    vals[idx] = list(observed_nodes.values())
    full, _, _ = conditional_gaussian_batch(np.asarray(model_uGLAD.precision_)[None], np.asarray(model_uGLAD.location_)[None],
                                            mask[None], vals[None], clip01=True)
    return full[0].cpu().numpy().astype(np.float64)


def save_uGLAD_model(obj, filepath: str) -> None:
    """Pickle the estimator's attributes to `filepath` and the 42 parameters (state_dict) to `filepath + "_model.pt"` -- the
    file layout of the reference (main.py:1132-1149)."""
    import pickle

    d = obj.__dict__.copy()
    has_model = d.get("model_glad") is not None
    if has_model:
        torch.save({k: v.detach().cpu() for k, v in obj.model_glad.state_dict().items()}, filepath + "_model.pt")
    d["model_glad"] = None
    d["_has_model"] = has_model  # (the reference tests the pickled None here and therefore never restores its model)
    with open(filepath, "wb") as f:
        pickle.dump(d, f)


def load_uGLAD_model(filepath: str):
    """Inverse of save_uGLAD_model (ref main.py:1152-1173, with its restore actually taking place): the estimator comes back
    with a GladParams on the current device, ready for predict()."""
    import pickle

    with open(filepath, "rb") as f:
        d = pickle.load(f)
    has_model = d.pop("_has_model", False)
    obj = uGLAD_multitask() if isinstance(d.get("covariance_"), list) or np.ndim(d.get("precision_")) == 3 else uGLAD_GL()
    obj.__dict__.update(d)
    if has_model:
        model = GladParams(1.0, device=_lib.device())
        model.load_state_dict(torch.load(filepath + "_model.pt", map_location="cpu"))
        obj.model_glad = model
    return obj
