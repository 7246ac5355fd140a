#!/usr/bin/env python3
"""Does a caller's hipGraph pay on the matrix-iteration path (some 60 short launches per step)?  One pass (forward + backward through the C
entry points) eager vs captured with torch.cuda.graph and replayed, L = 15.  python scripts/ns_graph_probe.py > gpurun_out/ns_graph_probe.txt"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uglad_amd import _lib
from uglad_amd.utils.prepare_data import synthetic_covariance_batch
lib = _lib.get_lib()
pz = np.load(os.path.join(ROOT, "tests", "golden", "params_trained.npz"))
pk = torch.tensor(np.concatenate([pz[k].ravel() for k in pz.files]), dtype=torch.float32, device="cuda")
f32 = dict(dtype=torch.float32, device="cuda")
L, mode = 15, _lib.SQRT_MODES["ns10"]
print(f"# ms per pass (forward + backward), L = {L}: eager / graph replay")
for D, M, forced in ((256, 1, 1), (256, 1, -1), (320, 1, -1), (512, 1, -1), (512, 4, -1), (128, 1024, -1)):
    lib.set_matrix_iteration(forced)
    S = torch.from_numpy(synthetic_covariance_batch(M, D, seed=D)).cuda()
    Z, half, U = torch.empty(L + 1, M, D, D, **f32), torch.empty(L, M, D, D, **f32), torch.empty(L, M, D, D, **f32)
    beta, lam, lam_in = torch.empty(L, M, D, **f32), torch.empty(L + 1, **f32), torch.empty(L + 1, 2, **f32)
    nfp, nfs, wsp = torch.empty(M, **f32), torch.empty(1, **f32), lib.workspace(M, D, S)
    GL = torch.randn(M, D, D, **f32); GL = (GL + GL.transpose(1, 2)).contiguous()
    gb0, gb1 = torch.empty(M, D, D, **f32), torch.empty(M, D, D, **f32)
    grp, glp, gtp, grad = torch.empty(M, 28, **f32), torch.empty(L, M, **f32), torch.empty(M, **f32), torch.empty(42, **f32)
    def one_pass():
        lib.glad_forward(S, pk, 1.0, 0, L, Z, half, U, beta, lam, lam_in, nfp, nfs, wsp, mode)
        lib.glad_backward(GL, S, pk, 0, L, Z, half, U, beta, lam, lam_in, gb0, gb1, grp, glp, gtp, grad, wsp, mode)
    def timeit(fn, reps=10):
        fn(); fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3
    eager = timeit(one_pass)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        one_pass(); torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            one_pass()
    rep = timeit(graph.replay)
    path = "matrix iteration" if (forced == 1 or D > 256) else "spectral"
    print(f"D={D:4d} M={M:5d} {path:17s} {eager:8.2f} / {rep:8.2f}", flush=True)
lib.set_matrix_iteration(-1)
