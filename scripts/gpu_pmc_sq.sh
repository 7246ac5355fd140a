#!/bin/bash
# Issue / stall / matrix-pipe counters of the hot kernels (rocprofv3 --pmc, one pass per counter group; program directly after
# `--`, no forked input generation under the counter collection).  Summary -> gpurun_out/pmc_sq_summary.txt
# Another configuration: PMC_M=1 PMC_D=25 PMC_L=15 PMC_STEPS=5 bash scripts/gpu_pmc_sq.sh
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ "${1:-}" = list ]; then rocprofv3 -L > gpurun_out/pmc_list.txt 2>&1; grep -c . gpurun_out/pmc_list.txt; exit 0; fi
GROUPS_=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
)
i=0
for g in "${GROUPS_[@]}"; do
  rm -rf gpurun_out/pmcsq_$i
  UGLAD_BENCH_NOFORK=1 timeout -k 10 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/pmcsq_$i -- python bench.py --steps ${PMC_STEPS:-1} --warmup 0 --no-cpu-baseline --M ${PMC_M:-1024} --D ${PMC_D:-128} --L ${PMC_L:-30} > gpurun_out/pmcsq_$i.log 2>&1
  rc=$?; echo "pmc group $i rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 90; fi
  i=$((i+1))
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/pmcsq_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void uglad::", "")
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
with open("gpurun_out/pmc_sq_summary.txt", "w") as fh:
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", [0, 1])[0])[:6]:
        line = f"== {k}  (per launch, summed over the chip)"
        print(line); fh.write(line + "\n")
        for c in sorted(agg[k]):
            v, n = agg[k][c]
            line = f"   {c:32s} {v / max(n, 1):16.1f}   ({n} launches)"
            print(line); fh.write(line + "\n")
PY
find gpurun_out/pmcsq_* -name "*.csv" -size +2M -delete 2>/dev/null
