#!/bin/bash
# Which side carries the trigger?  The SLP-packed narrow GroupNorm backward (victim as in arm A of gn_hazard.sh) beside a SYNTHETIC aggressor:
# fill loops of tools/probes/dma_writer_lib.hip in every combination of {LDS-DMA | load + ds_write} x {MFMAs} x {LDS reads}, 36 / 72 / 108 KB.
cd "$(dirname "$0")/../.." || exit 1
export EEGLDM_LIB=tools/debug/libeegldm_gnA.so EEGLDM_GN_BWD_NTH=256 NCALL=1
for kb in 36 108; do for mode in 1 3 7 0 2 6; do
  echo "== synthetic aggressor ${kb} KB mode $mode (1 = LDS-DMA, 2 = MFMA, 4 = LDS reads)"
  AGGRESSOR=synthetic:$kb:$mode timeout 200 python tools/debug/gn_hazard_diag.py 2>&1 | grep "^noisy" | sed 's/worst channels.*//'
done; done
echo "== the library's weight-gradient GEMM"; AGGRESSOR=wgrad timeout 200 python tools/debug/gn_hazard_diag.py 2>&1 | grep "^noisy" | sed 's/worst channels.*//'
