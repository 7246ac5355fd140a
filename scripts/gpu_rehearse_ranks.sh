#!/bin/bash
# Rehearsal of bench.py's multi-rank branch on a ONE-GPU box: two ranks share the device, the collectives ride on gloo (RCCL
# refuses two ranks on one device).  Proves that the N > 1 code path of bench.py (rendezvous, sharded inputs, the per-step
# scalar all-reduce, the 43-float gradient exchange, max-over-ranks timing) executes; it is not a scaling measurement.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
UGLAD_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --M ${REHEARSE_M:-256} --steps 3 --warmup 1 > gpurun_out/rehearse_2ranks.log 2>&1
echo "rc=$?"; tail -3 gpurun_out/rehearse_2ranks.log
