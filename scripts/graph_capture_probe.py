#!/usr/bin/env python3
"""Diagnostic (GPU box): capture one forward + backward pass into a caller-side graph and compare every buffer with plain launches."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uglad_amd
from uglad_amd import _lib
from oracle import glad_exact as ex
lib = _lib.get_lib()
g = np.load(os.path.join(ROOT, "tests", "golden", (sys.argv[1] if len(sys.argv) > 1 else "cell_d25_b1_L15_trained") + ".npz"))
m = uglad_amd.GladParams(1.0, device="cuda"); m.load_state_dict({k: torch.from_numpy(np.array(g["param." + k])) for k in ex.PARAM_KEYS})
pk = m.packed().detach().contiguous()
S = torch.from_numpy(g["S"]).cuda(); M, D, _ = S.shape; L = int(g["L"])
f32 = dict(dtype=torch.float32, device="cuda")
B = dict(Z=torch.empty(L + 1, M, D, D, **f32), half=torch.empty(L, M, D, D, **f32), U=torch.empty(L, M, D, D, **f32), beta=torch.empty(L, M, D, **f32),
         lam=torch.empty(L + 1, **f32), lam_in=torch.empty(L + 1, 2, **f32), nfp=torch.empty(M, **f32), nfs=torch.empty(1, **f32), cond=torch.empty(M, **f32),
         gb0=torch.empty(M, D, D, **f32), gb1=torch.empty(M, D, D, **f32), grp=torch.empty(M, 28, **f32), glp=torch.empty(L, M, **f32),
         gtp=torch.empty(M, **f32), grad=torch.empty(42, **f32))
wsp = lib.workspace(M, D, S)
GL = torch.randn(M, D, D, generator=torch.Generator(device="cuda").manual_seed(3), **f32); GL = (GL + GL.transpose(1, 2)).contiguous()
def one_pass(bwd=True):
    lib.glad_forward(S, pk, 1.0, 0, L, B["Z"], B["half"], B["U"], B["beta"], B["lam"], B["lam_in"], B["nfp"], B["nfs"], wsp, 1, cond_max=B["cond"])
    if bwd:
        lib.glad_backward(GL, S, pk, 0, L, B["Z"], B["half"], B["U"], B["beta"], B["lam"], B["lam_in"], B["gb0"], B["gb1"], B["grp"], B["glp"], B["gtp"], B["grad"], wsp, 1)
def snap():
    torch.cuda.synchronize(); return {k: v.clone() for k, v in B.items()}
def cmp(tag, a, b):
    bad = [f"{k}(nan {int(torch.isnan(a[k]).sum())}/{int(torch.isnan(b[k]).sum())})" for k in a if not torch.equal(a[k], b[k])]
    print(tag, "differs:", bad if bad else "nothing")
one_pass(); p1 = snap(); one_pass(); p2 = snap(); cmp("plain vs plain", p2, p1)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    one_pass(); s1 = snap(); cmp("side-stream plain vs default-stream plain", s1, p1)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        one_pass()
    gr.replay(); r1 = snap(); cmp("replay (on side) vs plain", r1, p1)
gr.replay(); r2 = snap(); cmp("replay (on default) vs plain", r2, p1)
for k in B: B[k].zero_()
gr.replay(); r3 = snap(); cmp("replay after zeroing vs plain", r3, p1)
gf = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(gf, stream=side):
        one_pass(bwd=False)
for k in B: B[k].zero_()
gf.replay(); r4 = snap(); cmp("forward-only replay after zeroing vs plain (forward buffers)", {k: r4[k] for k in ("Z","half","U","beta","lam","cond")}, p1)
