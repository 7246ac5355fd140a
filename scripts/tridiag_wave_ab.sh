#!/bin/bash
# Same-box A/B (GPU box) of the one-wave tridiagonalisation (tridiag_wave.h, D <= 64) against the workgroup kernel (UGLAD_TRIDIAG_WAVE=0) on
# bench.py's pass at several (M, D, L): ms per pass, forward-only rate and the HIP-event launch times, two alternating rounds.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/tridiag_wave_ab.txt
: > $OUT
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print(sys.argv[1], "ms/pass", d["ms_per_step"], "fwd-only steps/s", d["forward_only_steps_per_s"], " ".join(f"{k}={v['"'"'launch_ms'"'"']}" for k,v in r["other"].items()), "fwd_cell", r["forward_cell"]["launch_ms"])'
for cfg in "1 25 15" "128 64 30" "8 32 15" "2048 64 10" "4096 32 10" "1024 48 10"; do
  set -- $cfg
  for i in 1 2; do
    for mode in 1 0; do
      UGLAD_TRIDIAG_WAVE=$mode timeout -k 10 300 python bench.py --M $1 --D $2 --L $3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$pick" "M=$1 D=$2 L=$3 wave=$mode" | tee -a $OUT || exit 1
    done
  done
done
