#!/bin/bash
# Same-box A/B of two builds of the library on bench.py's workload: alternates UGLAD_LIB=<A> and the in-tree build, N rounds, prints ms/step and
# the per-kernel HIP-event launch times of every run.   bash scripts/ab_bench.sh scripts/_build/libuglad_r3.so [rounds=3] [bench args...]
set -u
cd "$(dirname "$0")/.."
A=$1; N=${2:-3}; shift; shift
mkdir -p gpurun_out
OUT=gpurun_out/ab_bench.txt
: > $OUT
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print(sys.argv[1], "ms/step", d["ms_per_step"], "fwd-only steps/s", d["forward_only_steps_per_s"], " ".join(f"{k}={v['"'"'launch_ms'"'"']}" for k,v in r["other"].items()), "fwd_cell", r["forward_cell"]["launch_ms"])'
for i in $(seq 1 $N); do
  UGLAD_LIB=$PWD/$A timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "$pick" "A($A)" | tee -a $OUT || exit 1
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "$pick" "B(in-tree)" | tee -a $OUT || exit 1
done
