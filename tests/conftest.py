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

    srcs = _kernel_sources()
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


# The AddressSanitizer build of the same sources (tests/test_kernels_sanitized.py): kept in-tree like the emulator library (git-ignored, rebuilt when a
# source is newer), and when both are stale its compile runs in the BACKGROUND from the end of collection, next to the emulator build the first
# emulated test triggers -- the two compiles are 5 of the suite's 9 minutes when run one after the other.
ASAN_LIB = os.path.join(EMUL_DIR, "libuglad_emul_asan.so")
_asan_build = {"proc": None}


def _kernel_sources():
    import glob

    return glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + [
        os.path.join(EMUL_DIR, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "uglad_hip.h")]


def asan_lib_is_fresh():
    return os.path.exists(ASAN_LIB) and all(os.path.getmtime(ASAN_LIB) >= os.path.getmtime(s) for s in _kernel_sources())


def start_asan_build():
    """Start the sanitizer compile (no-op when fresh, running, or without a host clang++); asan_lib() waits for it."""
    import subprocess

    if asan_lib_is_fresh() or _asan_build["proc"] is not None or not os.path.exists(HOST_CLANG):
        return
    # address + array-bounds only, line tables only, and only the padded sizes the script uses (NT = 1, 2, 5: UGLAD_NT_MASK): the full
    # UBSan + -g build of all instantiations takes 4 minutes
    cmd = [HOST_CLANG, "-x", "c++", "-std=c++17", "-O1", "-gline-tables-only", "-fPIC", "-shared", "-Wno-psabi", "-Wno-pass-failed", "-DUGLAD_MAX_NT=5",
           "-DUGLAD_NT_MASK=0x26", "-fsanitize=address,bounds", "-fno-sanitize-recover=bounds", "-shared-libasan", "-I", EMUL_DIR,
           os.path.join(CSRC, "glad_kernels.hip"), "-o", ASAN_LIB + ".tmp"]
    _asan_build["proc"] = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def asan_lib():
    """Path of the sanitizer build (compiling it now if nobody has), or None without a host clang++."""
    if asan_lib_is_fresh():
        return ASAN_LIB
    start_asan_build()
    proc = _asan_build["proc"]
    if proc is None:
        return None
    out, _ = proc.communicate()
    _asan_build["proc"] = None
    if proc.returncode != 0:
        raise RuntimeError("sanitizer build failed:\n" + (out or "")[-3000:])
    os.replace(ASAN_LIB + ".tmp", ASAN_LIB)
    return ASAN_LIB


def pytest_collection_finish(session):
    if any(item.name == "test_kernels_under_asan" for item in session.items):
        start_asan_build()


def pytest_sessionfinish(session, exitstatus):
    proc = _asan_build["proc"]
    if proc is not None and proc.poll() is None:  # (the run ended before the sanitizer test was reached, e.g. -x)
        proc.kill()
        proc.communicate()


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
