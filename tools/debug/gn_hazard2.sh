#!/bin/bash
# follow-up arms: the aggressor (weight-gradient GEMM) with extra dynamic LDS behind its tiles (EEGLDM_GEMM_LDS_PAD) -- do writes past the end of its allocation cause it?
cd "$(dirname "$0")/../.." || exit 1
for pad in 0 1024 8192 32768; do
  EEGLDM_GEMM_LDS_PAD=$pad EEGLDM_LIB=tools/debug/libeegldm_gnA.so EEGLDM_GN_BWD_NTH=256 timeout 300 python tools/debug/gn_hazard.py "A pad=$pad" 256 2>&1 | grep "^hazard.*wgrad"
  EEGLDM_GEMM_LDS_PAD=$pad EEGLDM_LIB=tools/debug/libeegldm_gnB.so EEGLDM_GN_BWD_NTH=256 timeout 300 python tools/debug/gn_hazard.py "B pad=$pad" 256 2>&1 | grep "^hazard.*wgrad"
done
for sw in EEGLDM_WGRAD_NO_DMA EEGLDM_GEMM_FUSED3_ATOMIC EEGLDM_NO_FUSED_BIAS_GRAD EEGLDM_GEMM_NO_SPLITK_WS; do
  env $sw=1 EEGLDM_LIB=tools/debug/libeegldm_gnB.so EEGLDM_GN_BWD_NTH=256 timeout 300 python tools/debug/gn_hazard.py "B $sw" 256 2>&1 | grep "^hazard.*wgrad"
done
