#!/bin/bash
# SQ counters over one LDM step per kernel class: wait / issue / LDS / MFMA shares
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_s21; rm -rf $OUT; mkdir -p $OUT
i=0
for G in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  i=$((i+1))
  EEGLDM_NO_SIDE_STREAM=1 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/g$i -o pmc -- python tools/debug/quick_bench.py bfloat16 256 768 2 > $OUT/g$i.log 2>&1
  find $OUT/g$i -name "*counter_collection.csv" -exec cp {} $OUT/cc$i.csv \;
  rm -rf $OUT/g$i
done
python - <<'P'
import csv, collections, glob, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('gpurun_out/r05_s21/cc*.csv')):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); k = re.sub(r'^void ', '', k); k = k.split('(')[0][:70]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
rows = []
for k, v in agg.items():
    tot = {c: sum(x) for c, x in v.items()}
    if tot.get('SQ_BUSY_CU_CYCLES', 0) < 5e7: continue
    rows.append((tot['SQ_BUSY_CU_CYCLES'], k, tot, len(v['SQ_BUSY_CU_CYCLES'])))
print("%-72s %5s %8s | of wave cycles: %6s %6s %6s %6s | %6s %6s" % ("kernel", "n", "busyM", "valu", "waitI", "waitM", "lds", "mfma%", "ldsarr%"))
for b, k, t, n in sorted(rows, reverse=True)[:24]:
    w = t['SQ_WAVE_CYCLES']
    print("%-72s %5d %8.1f | %6.1f %6.1f %6.1f %6.1f | %6.1f %6.1f" % (k, n, b / 1e6, 100 * t['SQ_ACTIVE_INST_VALU'] / w, 100 * t['SQ_WAIT_INST_ANY'] / w, 100 * t['SQ_WAIT_ANY'] / w,
          100 * t['SQ_ACTIVE_INST_LDS'] / w, 100 * t['SQ_VALU_MFMA_BUSY_CYCLES'] / (16 * b), 100 * t['SQ_LDS_IDX_ACTIVE'] / b))
P
