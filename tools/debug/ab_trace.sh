#!/bin/bash
# kernel-trace A/B of the LDM step: persistent big-tile kernels (default) vs EEGLDM_GEMM_BIG_NO_PERSIST=1; prints the big-tile rows
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp EEGLDM_NO_SIDE_STREAM=1
for v in P N; do
  if [ $v = N ]; then export EEGLDM_GEMM_BIG_NO_PERSIST=1; else unset EEGLDM_GEMM_BIG_NO_PERSIST; fi
  O=gpurun_out/ab_trace_$v; rm -rf $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python bench.py --no-parts --no-cpu-baseline --no-roofline --steps 7 > $O.line 2> $O.log
  python - "$O" "$v" <<'PY'
import csv, glob, sys, collections
d, v = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(v, "total kernel ms", round(tot / 1e6, 2))
for r in rows:
    n = r["Name"]
    if "gemm_big" in n or "gn_" in n[:6] or "gemm_kernel" in n:
        if float(r["TotalDurationNs"]) / tot > 0.01: print(f'  {v} {float(r["TotalDurationNs"])/1e6:8.3f} ms {int(r["Calls"]):5d} x {float(r["AverageNs"])/1e3:8.2f} us  {n[:90]}')
PY
done
