#!/bin/bash
# Builds one library per ablation variant of gemm_big.hip (compile-time bits, see the file) next to the production objects, then times each:
#   bash tools/debug/gemm_big_ablate.sh build      (here, cross-compiling)
#   bash tools/debug/gemm_big_ablate.sh run        (on the GPU box)
cd "$(dirname "$0")/../.." || exit 1
PKG=synthetic-sleep-eeg-signal-generation-using-latent-diffusion-models_amd
VARIANTS="${VARIANTS:-0 8 1 2 4 3 6 7 15}"
if [ "$1" = build ]; then
  for b in $VARIANTS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Iinclude -DEEG_BIG_DBG=$b -c $PKG/csrc/gemm_big.hip -o /tmp/gemm_big_$b.o 2>/dev/null || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $PKG/csrc/build/*.o | grep -v gemm_big.o) /tmp/gemm_big_$b.o -o tools/debug/libeegldm_big$b.so || exit 1
  done
else
  for b in $VARIANTS; do EEGLDM_LIB=tools/debug/libeegldm_big$b.so timeout 120 python tools/debug/gemm_big_ablate.py 2>&1 | grep "^B="; done
fi
