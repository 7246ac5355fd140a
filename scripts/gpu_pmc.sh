#!/bin/bash
# HBM traffic of the hot kernels from the PMC counters (separate passes: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
# Training and inference launches of the forward cell differ in algorithmic bytes (a training launch also writes U, theta_half and beta), so the
# two flavours are collected in runs of their own (bench.py --phase train | infer) and reported separately.
for ph in train infer; do
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${c}_$ph
  # (no forked input generation under the counter collection)
  UGLAD_BENCH_NOFORK=1 timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${c}_$ph -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --phase $ph > gpurun_out/pmc_${c}_$ph.log 2>&1
  rc=$?; echo "pmc $c $ph rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 90; fi
done
done
python - <<'PY'
import csv, glob, collections
with open("gpurun_out/pmc_summary.txt", "w") as fh:
    for ph in ("train", "infer"):
        out = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            files = glob.glob(f"gpurun_out/pmc_{c}_{ph}/**/*counter_collection.csv", recursive=True)
            agg = collections.defaultdict(lambda: [0.0, 0])
            for f in files:
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") == c:
                        k = r["Kernel_Name"].split("(")[0]
                        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
            out[c] = agg
        head = f"## {'training step (forward with saved state + backward)' if ph == 'train' else 'forward-only passes (inference)'}"
        print(head); fh.write(head + "\n")
        for k in sorted(out["FETCH_SIZE"], key=lambda k: -out["FETCH_SIZE"][k][0])[:8]:
            f, n = out["FETCH_SIZE"][k]; w, nw = out["WRITE_SIZE"].get(k, [0.0, 1])
            # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request: double it (MI355X_MICROARCH.md, HBM)
            line = f"{k[:60]:60s} launches {n:4d}  FETCH_SIZE/launch {f/n:12.1f} KiB (x2 corrected {2*f/n/1024:9.2f} MiB)  WRITE_SIZE/launch {w/max(nw,1)/1024:9.2f} MiB"
            print(line); fh.write(line + "\n")
PY
find gpurun_out/pmc_*SIZE_* -name "*.csv" -size +2M -delete 2>/dev/null
