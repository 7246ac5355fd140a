#!/bin/bash
# Instruction-fetch counters of the hot kernels (the forward cell's second stage is 90 KB of mostly straight-line code, the instruction
# cache 64 KB shared by two CUs): one rocprofv3 --pmc pass.  Summary -> gpurun_out/pmc_icache_summary.txt
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/pmcic
UGLAD_BENCH_NOFORK=1 timeout -k 10 300 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_WAVE_CYCLES \
  --kernel-trace --output-format csv -d gpurun_out/pmcic -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --M ${PMC_M:-1024} > gpurun_out/pmcic.log 2>&1
rc=$?; echo "pmc icache rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 90; fi
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/pmcic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void uglad::", "")
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
with open("gpurun_out/pmc_icache_summary.txt", "w") as fh:
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", [0, 1])[0])[:6]:
        line = f"== {k}  (per launch, summed over the chip)"
        print(line); fh.write(line + "\n")
        for c in sorted(agg[k]):
            v, n = agg[k][c]
            line = f"   {c:32s} {v / max(n, 1):16.1f}   ({n} launches)"
            print(line); fh.write(line + "\n")
PY
find gpurun_out/pmcic -name "*.csv" -size +2M -delete 2>/dev/null
