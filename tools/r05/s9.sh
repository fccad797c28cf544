#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_s9; rm -rf $OUT; mkdir -p $OUT
for G in 0 1; do EEGLDM_SAMPLE_GRAPH=$G python tools/debug/b1_trace.py 1 2>&1 | tail -2 | sed "s/^/graph=$G /"; done | tee $OUT/b1.log
EEGLDM_SAMPLE_GRAPH=1 EEGLDM_SAMPLE_NO_EMB_TABLE=1 python tools/debug/b1_trace.py 1 2>&1 | tail -1 | sed "s/^/graph=1 notable /" | tee -a $OUT/b1.log
EEGLDM_SAMPLE_OWN_STREAM=1 python tools/debug/b1_trace.py 1 2>&1 | tail -1 | sed "s/^/eager ownstream /" | tee -a $OUT/b1.log
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/t_full.log 2>&1; tail -6 $OUT/t_full.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r05_s9/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], {k:v for k,v in d['roofline'].items() if k.startswith('class')})
for k,v in d['parts'].items(): print(k, {kk:vv for kk,vv in v.items() if kk in ('windows_per_s','ms_per_step','batch1_latency_ms')})
P
