#!/bin/bash
# The GPU parity suite once per A/B switch of the library (GPU box): every alternative kernel choice the environment can select stays a tested path.
# One pytest process at a time; a run that times out ends the script.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/toggle_matrix.txt
: > $OUT
for t in "UGLAD_TRIDIAG_WAVE=0" "UGLAD_NO_FUSED_LAMBDA=1" "UGLAD_TRIDIAG_SMALL=0" "UGLAD_CHOLESKY=0" "UGLAD_PERSISTENT_BWD=0" "UGLAD_NS_PREFETCH_ALL=0" "UGLAD_LDL_LAUNCHES=0" "UGLAD_LDL_LAUNCHES=1" "UGLAD_NS_TILE=64"; do
  echo "=== $t ($(date +%T))"
  env $t timeout -k 10 500 python -m pytest tests -m gpu -q -x > gpurun_out/toggle_${t%%=*}_${t##*=}.log 2>&1
  rc=$?
  echo "$t rc=$rc $(tail -1 gpurun_out/toggle_${t%%=*}_${t##*=}.log)" | tee -a $OUT
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping" | tee -a $OUT; exit 90; fi
done
