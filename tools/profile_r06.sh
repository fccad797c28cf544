#!/bin/bash
# Collects the round-6 rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the headline bench command and of each secondary workload
#   2. PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, --kernel-trace only) of the LDM step and of the AEKL/GAN step
#   3. MFMA-busy pass of the LDM step
# Outputs land in gpurun_out/prof_r06/; tools/pmc_traffic.py + tools/prof_summary.py turn them into the files copied to profiles/.
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
export EEGLDM_PROFILE_ROUND=r06
OUT=gpurun_out/prof_r06; rm -rf $OUT; mkdir -p $OUT
TAG=${1:-v1}
MODE=${2:-full}      # "quick": kernel trace of the LDM step only
export EEGLDM_NO_SIDE_STREAM=1          # one stream: a launch's duration is its own
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ldm -o ldm -- python bench.py --no-parts --no-cpu-baseline --steps 7 > $OUT/bench_ldm_line.json 2> $OUT/trace_ldm.log
if [ "$MODE" = quick ]; then python tools/pmc_traffic.py $OUT $TAG trace-only > $OUT/summary.log 2>&1; tail -45 $OUT/summary.log; exit 0; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_aekl -o aekl -- python tools/debug/aekl_bench.py 256 bfloat16 > $OUT/aekl.txt 2> $OUT/trace_aekl.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_parts -o parts -- python tools/debug/parts_bench.py > $OUT/parts.txt 2> $OUT/trace_parts.log
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_ldm_$C -o pmc -- python tools/debug/quick_bench.py bfloat16 256 768 2 > $OUT/pmc_ldm_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_aekl_$C -o pmc -- python tools/debug/aekl_bench.py 256 bfloat16 > $OUT/pmc_aekl_$C.log 2>&1
done
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/pmc_ldm_mfma -o pmc -- python tools/debug/quick_bench.py bfloat16 256 768 2 > $OUT/pmc_ldm_mfma.log 2>&1
python tools/pmc_traffic.py $OUT $TAG > $OUT/summary.log 2>&1
tail -40 $OUT/summary.log
