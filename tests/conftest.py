import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---------------------------------------------------------------------------------------------------------------------
# Host build of the UNMODIFIED kernel sources on the SIMT emulator (tests/simt_emul): lets the CPU suite execute the real
# kernels' indexing / barrier / MFMA-lane logic.  Test infrastructure only -- the package never loads this library.
EMUL_DIR = os.path.join(ROOT, "tests", "simt_emul")
EMUL_LIB = os.path.join(EMUL_DIR, "libuglad_emul.so")
CSRC = os.path.join(ROOT, "uglad_amd", "csrc")
HOST_CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build_emulated_lib():
    import glob
    import subprocess

    srcs = glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + [
        os.path.join(EMUL_DIR, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "uglad_hip.h")]
    if os.path.exists(EMUL_LIB) and all(os.path.getmtime(EMUL_LIB) >= os.path.getmtime(s) for s in srcs):
        return EMUL_LIB
    if not os.path.exists(HOST_CLANG):
        return None
    # NT = 1, 2, 4, 5 only (D <= 64, 97..160: the sizes the CPU tests use, incl. the first workspace-resident one); the other
    # instantiations are GPU-tested
    cmd = [HOST_CLANG, "-x", "c++", "-std=c++17", "-O2", "-g", "-fPIC", "-shared", "-Wno-psabi", "-DUGLAD_MAX_NT=5", "-DUGLAD_NT_MASK=0x36", "-I", EMUL_DIR,
           os.path.join(CSRC, "glad_kernels.hip"), "-o", EMUL_LIB + ".tmp"]
    subprocess.run(cmd, check=True)
    os.replace(EMUL_LIB + ".tmp", EMUL_LIB)
    return EMUL_LIB


def install_emulated_lib():
    """Point uglad_amd at the host build (CPU tensors).  Returns the HipLib or None when no host clang++ is present."""
    import torch

    from uglad_amd import _lib

    path = build_emulated_lib()
    if path is None:
        return None
    _lib._instance = _lib.HipLib(path, require_gpu=False)
    _lib.device = lambda: torch.device("cpu")
    return _lib._instance


@pytest.fixture()
def emul():
    from uglad_amd import _lib

    saved = (_lib._instance, _lib.device)
    lib = install_emulated_lib()
    if lib is None:
        pytest.skip("host clang++ not available for the SIMT-emulator build")
    yield lib
    _lib._instance, _lib.device = saved
