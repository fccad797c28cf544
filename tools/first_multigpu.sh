#!/bin/bash
# First run on a node with >= 2 MI355X (the builder never had one: RCCL with more than one rank has not executed yet).
# Runs, in order of increasing blast radius, and stops at the first failure:
#   1. the two-rank RCCL tests (torch.distributed "nccl" AND the C-ABI communicator eegldm_comm_*), one GPU per rank
#   2. bench.py --gpus 2 / 4 / 8 on the default path (torch.distributed, ReduceOp.AVG, all-reduce overlapped with the backward)
#   3. the same with EEGLDM_NATIVE_COLLECTIVES=1 (collectives behind the C ABI)
# and prints, per run: collective_ranks / backend (a REAL all-reduce spanning all ranks), the bare all-reduce time of the 122 MB gradient
# buffer with its bus bandwidth, and the exposed communication per step.   Usage: bash tools/first_multigpu.sh [max_gpus]
set -u
cd "$(dirname "$0")/.." || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
NG=$(python - <<'P'
import torch; print(torch.cuda.device_count())
P
)
MAX=${1:-$NG}
echo "visible GPUs: $NG (using up to $MAX)"
if [ "$NG" -lt 2 ]; then echo "needs at least two GPUs"; exit 2; fi
python -m pytest tests/test_gpu_distributed.py -x -q -k "rccl_two_ranks_one_gpu_each" || { echo "FAILED: two-rank RCCL tests"; exit 1; }
summarise() {
python - "$1" <<'P'
import json, sys
line = [l for l in open(sys.argv[1]) if l.lstrip().startswith("{")][-1]
d = json.loads(line); c = d.get("comm") or d.get("config", {}).get("comm") or {}
keys = ("collective_ranks", "backend", "native_collectives", "allreduce_ms_bare", "allreduce_busbw_GBs", "compute_only_ms_per_step", "exposed_comm_ms_per_step")      # the keys of bench.py's `comm` object
print(f"  n_gpus {d['n_gpus']}: {d['value']:.0f} {d['unit']}, {d['ms_per_step']:.2f} ms/step | " + ", ".join(f"{k} {c.get(k, d.get(k))}" for k in keys))
P
}
for MODE in 0 1; do
  for N in 2 4 8; do
    [ "$N" -le "$MAX" ] || continue
    echo "== bench.py --gpus $N  EEGLDM_NATIVE_COLLECTIVES=$MODE"
    EEGLDM_NATIVE_COLLECTIVES=$MODE timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N + 10 * MODE)) \
      bench.py --gpus $N --steps 20 --warmup 5 --no-parts --no-cpu-baseline > /tmp/first_multigpu_${MODE}_$N.log 2>&1 || { tail -20 /tmp/first_multigpu_${MODE}_$N.log; echo "FAILED: bench --gpus $N mode $MODE"; exit 1; }
    summarise /tmp/first_multigpu_${MODE}_$N.log
  done
done
echo "all multi-GPU checks passed"
