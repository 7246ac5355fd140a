#!/bin/bash
# Same-box A/B (GPU box) of the lambda step carried by the cell's second launch when a group holds one matrix (LamStep,
# glad_kernels.hip) against its own norm_lambda launch (UGLAD_NO_FUSED_LAMBDA=1): BASELINE config 1, medians, three alternating rounds.
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  timeout -k 10 120 python scripts/c1_threads_probe.py || exit 90
  UGLAD_NO_FUSED_LAMBDA=1 timeout -k 10 120 python scripts/c1_threads_probe.py || exit 90
done
