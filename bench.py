#!/usr/bin/env python3
"""Benchmark of the unrolled-GLAD hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one training pass of the hot path over one batch of synthetic covariances already resident in HBM:
forward_uGLAD (Theta_0 init, L GLAD cells, glasso loss) + backward + gradient exchange + Adam.  Workload per GPU =
BASELINE.json config 3: M=1024 matrices, D=128, L=30, fp32; with N GPUs every rank holds its own 1024 (weak scaling,
global batch 1024*N = config 4 at N=8) and the per-step lambda scalar + the 43-float gradient message go over RCCL.
metric value = M_global * L * K / wall time (unroll-steps per second, whole job).

The JSON line also carries:
  roofline      the dominant kernel's ALGORITHMIC flops per launch (forward 20/3 D^3 + 50 D^2, backward
                8 D^3 + 100 D^2 per matrix) / its mean launch duration measured here with HIP events, against the f32 MFMA peak;
  cpu_baseline  the oracle's NS-faithful CPU restatement of the reference (oracle/glad_ns.py) timed on this host's cores on a
                bounded sub-batch of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters


def fwd_flops(D):
    return 20.0 / 3.0 * D**3 + 50.0 * D**2


def bwd_flops(D):
    return 8.0 * D**3 + 100.0 * D**2


def _gen_chunk(a):
    from uglad_amd.utils.prepare_data import synthetic_covariance_batch

    n, D, off = a
    try:  # one BLAS thread per worker process, or the workers thrash each other
        from threadpoolctl import threadpool_limits

        with threadpool_limits(limits=1):
            return synthetic_covariance_batch(n, D, seed=1234, task_offset=off)
    except ImportError:
        return synthetic_covariance_batch(n, D, seed=1234, task_offset=off)


def _host_cores() -> int:
    """Cores this process may really use: the affinity mask, capped at the GPU box's per-GPU CPU share (16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def _log(msg):
    print(f"[bench +{time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.time()


def _generate_inputs(M, D, offset):
    import multiprocessing as mp

    nw = max(1, min(8, _host_cores() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))
    chunks, lo = [], 0
    per = (M + nw * 4 - 1) // (nw * 4)
    while lo < M:
        n = min(per, M - lo)
        chunks.append((n, D, offset + lo))
        lo += n
    if nw == 1 or M < 32 or os.environ.get("UGLAD_BENCH_NOFORK") == "1":
        # (UGLAD_BENCH_NOFORK: under `rocprofv3 --pmc` the profiler has initialised the GPU before main() runs, and a process
        # that holds a GPU context must not fork)
        parts = []
        for k, c in enumerate(chunks):
            parts.append(_gen_chunk(c))
            if k % 8 == 7:
                _log(f"generated {sum(len(q) for q in parts)} / {M} covariances in-process")
    else:
        # close() + join(), not the context manager: that one terminate()s the workers (SIGTERM), which under rocprofv3's
        # chained signal handler leaves an abort trace in the profiler log
        pool = mp.get_context("fork").Pool(nw)
        try:
            parts = pool.map(_gen_chunk, chunks)
        finally:
            pool.close()
            pool.join()
    return np.concatenate(parts, axis=0)


def _baseline_config(M, D, L):
    """Which BASELINE.json configuration a (matrices per GPU, D, L) workload is, for the `config.workload` label."""
    if (D, L) == (128, 30):
        return "BASELINE config 3" if M == 1024 else ("BASELINE config 4's workload on one GPU" if M == 8192 else "config 3's shape, other batch")
    if (D, L) == (256, 30):
        return "BASELINE config 5's shape" + (": one matrix per GPU" if M == 1 else "")
    if (M, D, L) == (128, 64, 30):
        return "BASELINE config 2"
    if (M, D, L) == (1, 25, 15):
        return "BASELINE config 1"
    return "not a BASELINE configuration"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--M", type=int, default=1024, help="matrices per GPU")
    ap.add_argument("--D", type=int, default=128)
    ap.add_argument("--L", type=int, default=30)
    ap.add_argument("--sqrt-mode", default="ns10")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--phase", default="all", choices=("all", "train", "infer"),
                    help="counter collection only (scripts/gpu_pmc.sh): run just the training steps or just the forward-only passes, so that "
                         "per-launch HBM traffic can be reported per flavour (a training launch also writes U and theta_half); no JSON line")
    ap.add_argument("--cpu-sample", type=int, default=32, help="matrices in the CPU baseline sub-batch")
    ap.add_argument("--cpu-passes", type=int, default=6, help="timed CPU passes (plus one untimed)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    M, D, L = args.M, args.D, args.L
    Mg = M * world

    # ---- inputs (host, BEFORE the GPU is touched so that worker processes can be forked): M sampled Gaussian-graph
    # covariances per rank from the generator of SURVEY.md section 8d; task i of the global batch uses default_rng(1234 + i).
    t0 = time.time()
    S_host = _generate_inputs(M, D, rank * M)
    gen_s = time.time() - t0
    if rank == 0:
        _log(f"generated {M} x {D}x{D} covariances per rank in {gen_s:.1f} s ({_host_cores()} host cores usable)")
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    local_dev = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" IS RCCL on ROCm.  UGLAD_DIST_BACKEND=gloo exists only to rehearse the multi-rank path on a one-GPU box
        # (RCCL refuses two ranks on one device); local ranks then share the visible devices round-robin.
        backend = os.environ.get("UGLAD_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import uglad_amd
    from uglad_amd import _lib, main as um
    from uglad_amd.dist import get_collective
    lib = _lib.get_lib()
    coll = get_collective()
    # exchange (i), the per-step scalar: issued from the library over a communicator of its own when the group runs on RCCL
    # (uglad_glad_forward_sharded); set up here so that no timed step pays for ncclCommInitRank
    native = getattr(coll, "native_exchange", lambda: None)()
    exchange = "none (single process)" if world == 1 else ("ncclAllReduce issued from libuglad_hip.so on the compute stream" if native
                                                           else "torch.distributed.all_reduce per step")
    rccl = None
    if world > 1:
        # what a multi-GPU record needs to show that RCCL saw every rank: the library's own communicator (ncclCommCount) when the native
        # exchange is on, and in any case the sum of ones over the process group the per-step exchange and the gradient message use
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        rccl = dict(getattr(coll, "native_status", lambda: {"native": False, "nranks": None, "fallback_reason": "not a TorchCollective"})())
        rccl.update(backend=torch.distributed.get_backend(), process_group_allreduce_of_ones=float(ones.item()), world_size=world)
    S = torch.from_numpy(S_host).to(dev).contiguous()  # resident in HBM before any timing

    pz = np.load(os.path.join(ROOT, "tests", "golden", "params_trained.npz"))
    model = uglad_amd.GladParams(1.0, device=dev)
    model.load_state_dict({k: torch.from_numpy(np.array(pz[k])) for k in pz.files})
    opt = uglad_amd.get_optimizers(model, lr_glad=0.002)

    def train_step():
        opt.zero_grad()
        theta, loss = um.forward_uGLAD(S, model, L=L, sqrt_mode=args.sqrt_mode, collective=coll, global_batch=Mg)
        loss.backward()
        um._allreduce_grads(model, loss, coll)
        opt.step()
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.phase == "infer":  # (counter collection: forward-only launches alone)
        with torch.no_grad():
            for _ in range(max(1, args.steps)):
                um.forward_uGLAD(S, model, L=L, sqrt_mode=args.sqrt_mode, collective=coll, global_batch=Mg)
        barrier()
        return
    for _ in range(args.warmup):
        train_step()
    barrier()
    if rank == 0:
        _log("warm-up done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    value = Mg * L * args.steps / dt
    final_loss = float(loss.item())
    if rank == 0:
        _log(f"timed {args.steps} training steps: {dt / args.steps * 1e3:.1f} ms/step, {value:.0f} unroll-steps/s")
    if args.phase == "train":  # (counter collection: training launches alone)
        return

    # ---- forward-only rate (no_grad: the predict / CV-final path)
    with torch.no_grad():
        um.forward_uGLAD(S, model, L=L, sqrt_mode=args.sqrt_mode, collective=coll, global_batch=Mg)
        barrier()
        t0 = time.perf_counter()
        for _ in range(max(1, args.steps)):
            um.forward_uGLAD(S, model, L=L, sqrt_mode=args.sqrt_mode, collective=coll, global_batch=Mg)
        barrier()
        fwd_rate = Mg * L * max(1, args.steps) / (time.perf_counter() - t0)
    if rank == 0:
        _log(f"forward-only: {fwd_rate:.0f} unroll-steps/s")

    # ---- roofline of the dominant kernels: HIP events on the launch stream around single launches
    roof = None
    if rank == 0:
        mode = _lib.SQRT_MODES[args.sqrt_mode]
        f32 = dict(dtype=torch.float32, device=dev)
        pk = model.packed().detach().contiguous()
        Z0, Z1, half, U = (torch.empty(M, D, D, **f32) for _ in range(4))
        beta, nfp = torch.empty(M, D, **f32), torch.empty(M, **f32)
        lam, lam_in = torch.empty(2, **f32), torch.empty(2, 2, **f32)
        wsp = lib.workspace(M, D, S)
        lib.init_theta(S, pk, 0, Z0, wsp)
        lib.lambda_init(pk, 1.0, lam[0:1], lam_in[0])
        G0, G1 = torch.randn(M, D, D, **f32), torch.empty(M, D, D, **f32)
        G0 = (G0 + G0.transpose(1, 2)).contiguous()
        grp, glp = torch.zeros(M, 28, **f32), torch.empty(M, **f32)
        reps = 5

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a, b in evs:
                a.record()
                fn()
                b.record()
            torch.cuda.synchronize()
            return float(np.mean([a.elapsed_time(b) for a, b in evs])) * 1e-3

        t_f = timed(lambda: lib.cell_fwd(S, Z0, lam[0:1], pk, Z1, half, U, beta, nfp, wsp, mode))
        t_t = timed(lambda: lib.tridiagonalize(S, Z0, lam[0:1], Z1, wsp))

        def stage2():  # the second launch alone; its reflector scratch (= Z1) is consumed, so re-run stage 1 untimed first
            lib.cell_fwd_stage2(S, Z0, lam[0:1], pk, Z1, half, U, beta, nfp, wsp, mode)

        def timed_stage2():
            ts = []
            for _ in range(reps + 1):
                lib.tridiagonalize(S, Z0, lam[0:1], Z1, wsp)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                stage2()
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            return float(np.mean(ts[1:])) * 1e-3

        t_2 = timed_stage2()
        t_b = timed(lambda: lib.cell_bwd(G0, S, Z0, half, U, beta, lam[0:1], pk, G1, grp, glp, mode, workspace=wsp))
        bwd_note = "one launch per step (uglad_cell_bwd)"
        if D <= 128:
            # the backward pass of a training step is ONE launch over its L steps (uglad_glad_backward): a whole pass over the state of a
            # real forward pass, divided by L (includes Theta_0's gradient and the final reduction, 0.14 ms per pass at M = 1024)
            Zs, hs, Us = torch.empty(L + 1, M, D, D, **f32), torch.empty(L, M, D, D, **f32), torch.empty(L, M, D, D, **f32)
            bs, lams, lins = torch.empty(L, M, D, **f32), torch.empty(L + 1, **f32), torch.empty(L + 1, 2, **f32)
            nfs = torch.empty(1, **f32)
            lib.glad_forward(S, pk, 1.0, 0, L, Zs, hs, Us, bs, lams, lins, nfp, nfs, wsp, mode)
            g1b, glpb, gtp, grad = torch.empty(M, D, D, **f32), torch.empty(L, M, **f32), torch.empty(M, **f32), torch.empty(42, **f32)
            t_b = timed(lambda: lib.glad_backward(G0, S, pk, 0, L, Zs, hs, Us, bs, lams, lins, G1, g1b, grp, glpb, gtp, grad, wsp, mode)) / L
            bwd_note = f"1/{L} of the one-launch backward pass (uglad_glad_backward over the state of a real {L}-step forward pass)"
            del Zs, hs, Us, bs, g1b
        # uglad_cell_fwd = tridiag_kernel + cell_fwd_kernel back to back on one stream; the forward cell's algorithmic flops
        # (20/3 D^3 + 50 D^2) split as 4/3 D^3 (tridiagonalisation) + the rest (D&C, back-transform, U phi U^T, epilogue)
        tri_fl = 4.0 / 3.0 * D**3 * M
        fwd_name = "cell_fwd_lean_kernel"
        kern = [(fwd_name, t_2, fwd_flops(D) * M - tri_fl), ("tridiag_kernel", t_t, tri_fl),
                ("cell_bwd_kernel", t_b, bwd_flops(D) * M)]
        name, tk, fl = max(kern, key=lambda x: x[1])
        ach = fl / tk / 1e12
        # HBM bytes per launch come from separate rocprofv3 --pmc passes (scripts/gpu_pmc.sh); the committed summary is for
        # exactly this workload, so it is attached only then
        traffic, traffic_inference, counters, pmc_source = None, None, None, None
        import glob

        summaries = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_summary.json")))  # the newest round's
        pmc = summaries[-1] if summaries else ""
        if os.path.exists(pmc) and (M, D) == (1024, 128):
            # NOT measured in this run: counters need their own rocprofv3 --pmc passes; these come from the committed summary
            pmc_source = "profiles/" + os.path.basename(pmc) + " (separate rocprofv3 --pmc passes of this workload, scripts/gpu_pmc*.sh)"
            rec = json.load(open(pmc)).get(name)
            if rec:
                traffic = rec.get("hbm_bytes_per_launch")  # the TRAINING flavour of the launch (it also writes U, theta_half, beta): the one timed here
                traffic_inference = rec.get("hbm_bytes_per_launch_inference")
                counters = {k: rec[k] for k in ("mfma_pipe_busy_frac", "valu_busy_frac", "wave_wait_frac", "lds_bank_conflict_frac")
                            if k in rec}
        roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_flavour": "training launch (forward-only launch: traffic_inference)",
                "traffic_inference": traffic_inference, "pmc": counters, "pmc_source": pmc_source,
                "launch_ms": round(tk * 1e3, 3),
                "flops_per_launch": fl,
                "forward_cell": {"launch_ms": round(t_f * 1e3, 3), "achieved": round(fwd_flops(D) * M / t_f / 1e12, 3),
                                 "frac": round(fwd_flops(D) * M / t_f / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)},
                "other": {k: {"launch_ms": round(t * 1e3, 3), "achieved": round(f / t / 1e12, 3)} for k, t, f in kern},
                "cell_bwd_kernel_timing": bwd_note}

    # ---- CPU baseline: the oracle's NS-faithful restatement of the reference on a bounded sub-batch (rank 0, N=1)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import glad_ns as ns

        _log(f"roofline: {roof}")
        ncpu = _host_cores()
        torch.set_num_threads(ncpu)
        mc = min(args.cpu_sample, M)
        Sc = S[:mc].cpu()
        sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
        def cpu_pass():
            for v in sd.values():
                v.grad = None
            th, ls = ns.forward_uGLAD(Sc, sd, L=L)
            ls.backward()

        # parity spot-check of THIS run's inputs, outside every timed region: the first two matrices of the seeded batch alone (lambda_k depends
        # on the batch mean, so a sub-batch is its own problem on both sides) on the GPU and through the oracle
        S2 = S[:min(2, M)].contiguous()
        with torch.no_grad():
            th_gpu, _ = um.forward_uGLAD(S2, model, L=L, sqrt_mode=args.sqrt_mode)
            th_cpu, _ = ns.forward_uGLAD(S2.cpu(), {k: v.detach() for k, v in sd.items()}, L=L)
        spot = max(float(torch.linalg.norm(th_gpu[i].cpu().double() - th_cpu[i].double()) / torch.linalg.norm(th_cpu[i].double()))
                   for i in range(S2.shape[0]))
        _log(f"parity spot-check on the bench's own first {S2.shape[0]} matrices vs oracle/glad_ns.py: Theta rel-Frobenius {spot:.2e}")
        cpu_pass()  # untimed: thread-pool spin-up, allocator
        t0 = time.perf_counter()
        for _ in range(args.cpu_passes):
            cpu_pass()
        tc = time.perf_counter() - t0
        rate = mc * L * args.cpu_passes / tc
        _log(f"cpu baseline: {rate:.1f} unroll-steps/s on {ncpu} threads ({tc:.1f} s)")
        cpu = {"value": round(rate, 2), "unit": "unroll-steps/s", "cores": torch.get_num_threads(), "kind": "port",
               "parity_spot_check": {"what": f"Theta_L of the first {S2.shape[0]} matrices of this run's seeded batch, run alone: HIP path vs "
                                             "oracle/glad_ns.py (fp32 CPU restatement), max relative Frobenius; tolerance 1e-4",
                                     "theta_relF": float(f"{spot:.3e}"), "ok": bool(spot < 1e-4)},
               "sample": f"{args.cpu_passes} training passes (fwd+bwd) of oracle/glad_ns.py (batched-bmm restatement of the reference, "
                         f"a stronger baseline than its per-matrix Python loop) on the first {mc} of the {M} matrices, D={D}, L={L}, "
                         f"{tc:.1f} s in all; the path is linear in M, so steps/s carries over to the full batch"}

    if rank == 0:
        metric = "GLAD unroll-steps/sec (batch D\u00d7D \u0398-updates), M=1024 D=128 L=30"  # BASELINE.json's metric, verbatim
        try:
            with open(os.path.join(ROOT, "BASELINE.json")) as fh:
                metric = json.load(fh).get("metric", metric)
        except (OSError, ValueError):
            pass
        out = {
            "metric": metric,
            "value": round(value, 1),
            "unit": "unroll-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"multi-task M={M}/GPU D={D} L={L} fp32 ({_baseline_config(M, D, L)}; global batch {Mg})",
                       "pass": "training step: forward + loss + backward + gradient exchange + Adam",
                       "sqrt_mode": args.sqrt_mode, "parallelism": f"batch-sharded x{world}", "per_step_exchange": exchange},
            "forward_only_steps_per_s": round(fwd_rate, 1),
            "final_loss": final_loss,
            "input_gen_s": round(gen_s, 1),
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if rccl is not None:
            out["rccl"] = rccl
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
