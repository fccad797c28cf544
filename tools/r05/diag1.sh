#!/bin/bash
# Round-5 GPU session 1: where do the runtime copy / fill dispatches of a step come from (launch sequence of one step), baseline numbers.
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_diag1; rm -rf $OUT; mkdir -p $OUT
python -m pytest tests/test_gpu_autograd.py -x -q -k "intervening" > $OUT/t_tape.log 2>&1; tail -3 $OUT/t_tape.log
export EEGLDM_NO_SIDE_STREAM=1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_ldm -o ldm -- python tools/debug/quick_bench.py bfloat16 256 768 3 > $OUT/quick.log 2>&1
F=$(find $OUT/trace_ldm -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py "$F" adam_kernel --seq > $OUT/seq_ldm.txt 2>&1
grep -n -B2 -A1 "rocclr" $OUT/seq_ldm.txt | head -150 > $OUT/seq_ldm_copies.txt
head -60 $OUT/seq_ldm.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_aekl -o aekl -- python tools/debug/aekl_bench.py 256 bfloat16 > $OUT/aekl.log 2>&1
F=$(find $OUT/trace_aekl -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py "$F" adam_kernel --seq > $OUT/seq_aekl.txt 2>&1
unset EEGLDM_NO_SIDE_STREAM
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json
for G in 0 1; do EEGLDM_SAMPLE_GRAPH=$G python tools/debug/b1_trace.py 1 > $OUT/b1_graph$G.log 2>&1; tail -2 $OUT/b1_graph$G.log; done
rm -rf $OUT/trace_ldm $OUT/trace_aekl
