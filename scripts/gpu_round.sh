#!/bin/bash
# One gpurun call: smoke -> GPU parity tests -> bench -> rocprof kernel stats.  Every step runs under its own timeout; a
# step that times out or is killed ends the call (no further GPU work after a suspected hang).  Logs go to gpurun_out/.
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
step() {  # step <name> <timeout_s> <cmd...>
  local name=$1 to=$2; shift 2
  echo "=== $name ($(date +%T))"
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc"
  tail -n 15 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "!!! $name timed out -- stopping"; exit 90; fi
  return $rc
}
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
WHAT=${1:-all}
if [ "$WHAT" = all ] || [ "$WHAT" = smoke ]; then step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"; fi
if [ "$WHAT" = all ] || [ "$WHAT" = tests ]; then step pytest_gpu 900 python -m pytest tests -m gpu -q -s --durations=10; fi
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then step bench 600 python bench.py --steps ${BENCH_STEPS:-3} --warmup 1; fi
if [ "$WHAT" = all ] || [ "$WHAT" = prof ]; then
  rm -rf gpurun_out/prof
  UGLAD_BENCH_NOFORK=1 step rocprof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline
  find gpurun_out/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats.csv \; 2>/dev/null
  find gpurun_out/prof -name "*kernel_trace.csv" -exec rm {} \; 2>/dev/null   # per-dispatch rows: large
  head -20 gpurun_out/kernel_stats.csv 2>/dev/null
fi
exit 0
