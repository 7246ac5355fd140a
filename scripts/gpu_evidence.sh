#!/bin/bash
# One gpurun call that collects everything the round's measurement claims rest on (copied to profiles/ afterwards):
#   kernel stats (rocprofv3 --kernel-trace --stats), HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes), issue / stall /
#   matrix-pipe counters (scripts/gpu_pmc_sq.sh), the bench line with the CPU baseline, the config-4 workload on one GPU,
#   the two-rank rehearsal of bench.py's multi-rank branch, the other configurations.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export UGLAD_BENCH_NOFORK=1  # no forked input generation under rocprofv3 (its preloaded library initialises the GPU first)
O=gpurun_out/evidence; rm -rf $O; mkdir -p $O
run() { local name=$1 to=$2; shift 2; echo "=== $name"; timeout -k 10 "$to" "$@" > $O/$name.log 2>&1; local rc=$?; echo "=== $name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 90; fi; }
run bench 400 python bench.py --steps 20 --warmup 5
run bench_m8192 300 python bench.py --M 8192 --steps 3 --warmup 1 --no-cpu-baseline
rm -rf gpurun_out/prof
run rocprof 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
find gpurun_out/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \; 2>/dev/null
find gpurun_out/prof -name "*kernel_trace.csv" -delete 2>/dev/null
bash scripts/gpu_pmc.sh > $O/pmc_hbm.log 2>&1; cp gpurun_out/pmc_summary.txt $O/pmc_hbm_summary.txt 2>/dev/null
bash scripts/gpu_pmc_sq.sh > $O/pmc_sq.log 2>&1; cp gpurun_out/pmc_sq_summary.txt $O/ 2>/dev/null
bash scripts/gpu_rehearse_ranks.sh > /dev/null 2>&1; cp gpurun_out/rehearse_2ranks.log $O/ 2>/dev/null
run other_configs 400 python scripts/bench_configs.py
# per-kernel split of the small configurations (BASELINE configs 1 and 2)
for cfg in "c1 --M 1 --D 25 --L 15" "c2 --M 128 --D 64 --L 30"; do
  set -- $cfg; tag=$1; shift
  rm -rf gpurun_out/prof_$tag
  run rocprof_$tag 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py "$@" --steps 20 --warmup 2 --no-cpu-baseline
  find gpurun_out/prof_$tag -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$tag.csv \; 2>/dev/null
  find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete 2>/dev/null
done
tail -2 $O/bench.log | cut -c1-400
