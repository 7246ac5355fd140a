#!/bin/bash
# rocprofv3 kernel stats of bench.py at one small configuration (GPU box): bash scripts/rocprof_small.sh <tag> <M> <D> <L>; top rows to gpurun_out/rocprof_<tag>.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp UGLAD_BENCH_NOFORK=1
tag=$1; M=$2; D=$3; L=$4
rm -rf gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --M $M --D $D --L $L --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 || exit 90
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' > gpurun_out/rocprof_$tag.txt
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[:7]:
    print(r[0][:60].ljust(60), r[1:4], r[5:7])
PY
find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
cat gpurun_out/rocprof_$tag.txt
