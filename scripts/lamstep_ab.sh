#!/bin/bash
# Same-box A/B (GPU box) of the lambda step carried by the forward cell's second launch (LamStep, glad_kernels.hip: thread 0 when a group holds one
# matrix, the last workgroup of the group to arrive otherwise) against its own norm_lambda launch (UGLAD_NO_FUSED_LAMBDA=1).
#   bash scripts/lamstep_ab.sh c1     BASELINE config 1, medians of individually synchronised passes, three alternating rounds
#   bash scripts/lamstep_ab.sh bench  bench.py at config 3 (the headline) and config 2, three alternating rounds; final_loss must agree to the bit
cd "$(dirname "$0")/.."
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], "ms/pass", d["ms_per_step"], "steps/s", d["value"], "fwd-only steps/s", d["forward_only_steps_per_s"], "final_loss", repr(d.get("final_loss")))'
if [ "${1:-c1}" = c1 ]; then
  for r in 1 2 3; do
    timeout -k 10 120 python scripts/c1_threads_probe.py || exit 90
    UGLAD_NO_FUSED_LAMBDA=1 timeout -k 10 120 python scripts/c1_threads_probe.py || exit 90
  done
else
  for cfg in "1024 128 30" "128 64 30"; do
    set -- $cfg
    for r in 1 2 3; do
      timeout -k 10 300 python bench.py --M $1 --D $2 --L $3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$pick" "M=$1 D=$2 L=$3 in the cell     " || exit 90
      UGLAD_NO_FUSED_LAMBDA=1 timeout -k 10 300 python bench.py --M $1 --D $2 --L $3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$pick" "M=$1 D=$2 L=$3 launch of its own" || exit 90
    done
  done
fi
