#!/bin/bash
# Development builds of the library (never the shipped one), made in the build container so that the GPU box spends no time
# compiling: scripts/_build/libuglad_diag.so = single translation unit, -DUGLAD_STAMPS (phase stamps), NT <= 4.
# Extra -D flags: EXTRA="-DFOO=1" bash scripts/dev_build.sh [name]
set -e
cd "$(dirname "$0")/.."
mkdir -p scripts/_build
NAME=${1:-diag}
FLAGS="-DUGLAD_STAMPS"
if [ "$NAME" != diag ]; then FLAGS=""; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DUGLAD_MAX_NT=4 $FLAGS ${EXTRA:-} uglad_amd/csrc/glad_kernels.hip -o scripts/_build/libuglad_$NAME.so
ls -la scripts/_build/libuglad_$NAME.so
