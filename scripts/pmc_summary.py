#!/usr/bin/env python3
"""profiles/rNN_pmc_summary.json from the two counter summaries scripts/gpu_pmc.sh and scripts/gpu_pmc_sq.sh write
(pmc_hbm_summary.txt, pmc_sq_summary.txt): per kernel the derived figures bench.py quotes in `roofline`.

    python scripts/pmc_summary.py gpurun_out/evidence profiles/r02_pmc_summary.json

Derivations (MI355X: 256 CUs x 4 SIMDs, 8 XCDs; counters are sums over the chip per launch):
  kernel_cycles          GRBM_GUI_ACTIVE / 8                        (the counter is summed over the 8 XCDs)
  mfma_pipe_busy_frac    SQ_VALU_MFMA_BUSY_CYCLES / (kernel_cycles * 1024)
  valu_busy_frac         4 * SQ_ACTIVE_INST_VALU / (kernel_cycles * 1024)   (the counter ticks once per 4 cycles of a SIMD)
  wave_wait_frac         SQ_WAIT_ANY / SQ_WAVE_CYCLES
  lds_bank_conflict_frac SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  fetch_bytes            FETCH_SIZE (KiB) * 1024 * 2                 (the gfx950 correction of MI355X_MICROARCH.md)
  write_bytes            WRITE_SIZE
"""
import json
import re
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    out = {}
    cur = None
    raw = {}
    for line in open(f"{src}/pmc_sq_summary.txt"):
        if line.startswith("== "):  # (a section of a kernel that is not ours -- "== void at::native::..." -- ends the previous one)
            m = re.match(r"== (\w+)<", line)
            cur = m.group(1) if m else None
            if cur:
                raw.setdefault(cur, {})
            continue
        f = line.split()
        if cur and len(f) >= 2 and cur in raw and f[0] not in raw[cur]:
            raw[cur][f[0]] = float(f[1])
    for k, c in raw.items():
        if "GRBM_GUI_ACTIVE" not in c:
            continue
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        out[k] = {
            "kernel_cycles": round(cyc),
            "mfma_pipe_busy_frac": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024), 4),
            "valu_busy_frac": round(4 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (cyc * 1024), 4),
            "wave_wait_frac": round(c.get("SQ_WAIT_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0), 4),
            "lds_bank_conflict_frac": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0), 4),
            "mfma_insts": c.get("SQ_INSTS_MFMA", 0.0),
            "valu_insts": c.get("SQ_INSTS_VALU", 0.0),
        }
    # HBM traffic per flavour (scripts/gpu_pmc.sh, round 4): a training launch of the forward cell also writes U, theta_half and beta, so the
    # two flavours have different algorithmic bytes and are never averaged.  `hbm_bytes_per_launch` = the training flavour (what bench.py's
    # `value` is made of), `hbm_bytes_per_launch_inference` beside it.  (A summary without section headers -- rounds 1-3 -- is all "training".)
    flavour = "training"
    for line in open(f"{src}/pmc_hbm_summary.txt"):
        if line.startswith("## "):
            flavour = "inference" if "forward-only" in line else "training"
            continue
        m = re.match(r"void uglad::(\w+)<.*FETCH_SIZE/launch\s+([\d.]+) KiB.*WRITE_SIZE/launch\s+([\d.]+) MiB", line)
        if not m or m.group(1) not in out:
            continue
        fetch = float(m.group(2)) * 1024 * 2
        write = float(m.group(3)) * 1024 * 1024
        rec = out[m.group(1)]
        if flavour == "training" and "fetch_bytes" not in rec:
            rec.update(fetch_bytes=round(fetch, 2), write_bytes=round(write, 2), hbm_bytes_per_launch=round(fetch + write))
        elif flavour == "inference" and "fetch_bytes_inference" not in rec:
            rec.update(fetch_bytes_inference=round(fetch, 2), write_bytes_inference=round(write, 2),
                       hbm_bytes_per_launch_inference=round(fetch + write))
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    for k, v in out.items():
        print(k, v)


if __name__ == "__main__":
    main()
