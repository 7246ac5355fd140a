#!/bin/bash
# Register allocation of ONE kernel family in seconds: compiles the device side of the NT translation unit with only that family
# instantiated (-DUGLAD_DEV_ONLY_<FAMILY>) and prints the resource remarks; the assembly stays in /tmp/asm/<family><NT>.s.
#   bash scripts/spill_check.sh [NT=4] [family=LEAN|TRIDIAG|TRIWAVE|BWD|CHOL] [extra -D flags]
NT=${1:-4}; FAM=${2:-LEAN}; shift; shift
mkdir -p /tmp/asm
OUT=/tmp/asm/$(echo $FAM | tr A-Z a-z)$NT.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DUGLAD_MAX_NT=8 -DUGLAD_TU_NT=$NT -DUGLAD_DEV_ONLY_$FAM "$@" --cuda-device-only -S \
  "$(dirname "$0")/../uglad_amd/csrc/glad_kernels.hip" -o $OUT -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|VGPRs|Scratch|Spill|LDS Size|SGPRs:" | sed 's/.*remark: *//; s/ \[-Rpass.*//'
echo "scratch instructions: $(grep -c scratch_ $OUT)  ($OUT)"
