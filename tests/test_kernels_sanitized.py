"""CPU suite: the kernel sources on the SIMT emulator under AddressSanitizer + bounds checking (GPU sanitizers are not available on
the pool, so this is where out-of-bounds LDS / global indexing and use-after-free of caller buffers get caught).  Runs a
small forward + backward + consensus + symeig sweep -- including D = 129, the workspace-slab path, and the matrix-iteration path at
D = 33 and D = 162 -- in a subprocess with the sanitizer runtime preloaded."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

_SCRIPT = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, {root!r})
from uglad_amd import _lib
lib = _lib.HipLib({so!r}, require_gpu=False)
_lib._instance = lib
_lib.device = lambda: torch.device("cpu")
import uglad_amd
from oracle import glad_exact as ex
for D in (1, 5, 33, 129):  # 129: the first workspace-resident size (NT = 5: buffers in global slabs, not LDS)
    A = torch.randn(1, D, D); A = (A + A.transpose(1, 2)).contiguous()
    beta, U = uglad_amd.batch_symeig(A)
    rec = (U * beta[:, None, :]) @ U.transpose(1, 2)
    assert float((rec - A).norm() / A.norm().clamp_min(1e-30)) < 5e-6, D
g = np.load(os.path.join({root!r}, "tests", "golden", "cell_d20_b5_L15_trained.npz"))
m = uglad_amd.GladParams(1.0)
m.load_state_dict({{k: torch.from_numpy(np.array(g["param." + k])) for k in ex.PARAM_KEYS}})
theta, loss = uglad_amd.forward_uGLAD(torch.from_numpy(g["S"][:2].copy()), m, L=2)
loss.backward()
assert torch.isfinite(theta).all() and all(torch.isfinite(p.grad).all() for p in m.parameters())
g9 = np.load(os.path.join({root!r}, "tests", "golden", "cell_d129_b2_L30_trained.npz"))
lib.set_matrix_iteration(0)  # the spectral kernels at D = 129 (left alone, one such matrix goes to the matrix-iteration path: below)
for wide in (0, 1):  # one workgroup per matrix / many (csrc/wide_bwd.h: ragged 64 x 64 tiles at D = 129)
    lib.set_wide_mode(wide)
    th9, ls9 = uglad_amd.forward_uGLAD(torch.from_numpy(g9["S"][:1].copy()), m, L=1)
    ls9.backward()
    assert torch.isfinite(th9).all() and all(torch.isfinite(p.grad).all() for p in m.parameters())
lib.set_wide_mode(-1)
# the matrix-iteration path (csrc/wide_ns.h): forced at D = 33 (odd: scalar operand loads, ragged tiles) on both output tilings and
# at D = 34 (even: 16-byte operand loads); beyond this build's eigensolver, at D = 162, the padded L D L^T on 16 waves with its Newton
# steps and the loss's trace (the cell itself at that size costs two minutes under the sanitizer and indexes as at 33 / 34)
from uglad_amd.utils.prepare_data import synthetic_covariance_batch
for D, tile in ((33, "32"), (33, "64"), (34, "32")):
    os.environ["UGLAD_NS_TILE"] = tile
    lib.set_matrix_iteration(1)
    for p in m.parameters():
        p.grad = None
    thn, lsn = uglad_amd.forward_uGLAD(torch.from_numpy(synthetic_covariance_batch(1, D, seed=D)), m, L=1)
    lsn.backward()
    assert torch.isfinite(thn).all() and torch.isfinite(lsn) and all(torch.isfinite(p.grad).all() for p in m.parameters()), D
lib.set_matrix_iteration(-1)
th162 = torch.from_numpy(synthetic_covariance_batch(1, 162, seed=162)).requires_grad_(True)
ls162 = uglad_amd.loss_uGLAD(th162, torch.eye(162)[None].contiguous())
ls162.backward()
assert torch.isfinite(ls162) and torch.isfinite(th162.grad).all()
lib.set_matrix_iteration(-1)
out = uglad_amd.get_final_precision_from_batch(theta.detach(), type="min")
assert out.shape == (1, 20, 20)
print("SANITIZED-OK")
"""


def test_kernels_under_asan(tmp_path):
    if not os.path.exists(CLANG):
        pytest.skip("host clang++ not available")
    asan = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(asan):
        pytest.skip("ASan runtime not available")
    from conftest import asan_lib

    so = asan_lib()  # (in-tree, rebuilt when a kernel source is newer; its compile started in the background at the end of collection)
    script = tmp_path / "run.py"
    script.write_text(_SCRIPT.format(root=ROOT, so=so))
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "SANITIZED-OK" in r.stdout, tail
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, tail
