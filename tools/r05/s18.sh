#!/bin/bash
# SQ counters of the GEMM kernels at 512 -> 512, L = 192, B = 256 (gemm_bench.py bf16 one): where do the cycles of the weight gradient go?
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_s18; rm -rf $OUT; mkdir -p $OUT
i=0
for G in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
         "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" \
         "SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL" \
         "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_ANY" \
         "SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/g$i -o pmc -- python tools/debug/gemm_bench.py bf16 one > $OUT/g$i.log 2>&1
  find $OUT/g$i -name "*counter_collection.csv" -exec cp {} $OUT/cc$i.csv \;
  rm -rf $OUT/g$i
done
ls -la $OUT
