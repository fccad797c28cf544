#!/bin/bash
# SQ counters of the thin autoencoder's whole-network kernels inside the AEKL / GAN step (is thin_bwd issue-bound or latency-bound?)
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_s20; rm -rf $OUT; mkdir -p $OUT
i=0
for G in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
         "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/g$i -o pmc -- python tools/debug/aekl_bench.py 256 bfloat16 > $OUT/g$i.log 2>&1
  find $OUT/g$i -name "*counter_collection.csv" -exec cp {} $OUT/cc$i.csv \;
  rm -rf $OUT/g$i
done
python - <<'P'
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('gpurun_out/r05_s20/cc*.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        for n in ('thin_bwd_kernel', 'thin_fwd_kernel', 'bn_apply4_kernel', 'spectral_kernel'):
            if n in k: agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k, "launches", len(next(iter(v.values()))))
    for c in sorted(v): print("   %-28s %14.0f" % (c, sum(v[c]) / len(v[c])))
P
