#!/bin/bash
# Round 6, VERDICT r5 item 2: ablation builds of the GroupNorm-on-operand-load conv (gemm_big_kernel<XF = 1>).
#   bash tools/r06/gn_onload_ablate.sh build     (cross-compile here: one library per variant under tools/r06/)
#   bash tools/r06/gn_onload_ablate.sh run       (on the GPU box: the bench of each variant; only the timing of the fused line is meaningful for variants 1-3)
cd "$(dirname "$0")/../.." || exit 1
PKG=synthetic-sleep-eeg-signal-generation-using-latent-diffusion-models_amd
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -fno-slp-vectorize -fno-vectorize -Iinclude"
declare -A V=( [novalu]="-DEEG_BIG_XF_ABL=1" [noxform]="-DEEG_BIG_XF_ABL=2" [nomfma]="-DEEG_BIG_DBG=4" [nomfma_noxform]="-DEEG_BIG_DBG=4 -DEEG_BIG_XF_ABL=2" )
if [ "$1" = build ]; then
  for k in "${!V[@]}"; do
    /opt/rocm/bin/hipcc $FLAGS ${V[$k]} -c $PKG/csrc/gemm_big.hip -o /tmp/gemm_big_$k.o 2>/dev/null || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $PKG/csrc/build/*.o | grep -v gemm_big.o) /tmp/gemm_big_$k.o -o tools/r06/libeegldm_xf_$k.so || exit 1
  done
else
  echo "== as built"; timeout 200 python tools/r06/gn_onload_bench.py $2 $3 $4 $5 2>&1 | grep -v amdgpu
  for k in novalu noxform nomfma nomfma_noxform; do
    echo "== $k (${V[$k]})"; EEGLDM_LIB=tools/r06/libeegldm_xf_$k.so timeout 200 python tools/r06/gn_onload_bench.py $2 $3 $4 $5 2>&1 | grep "fused\|one tile per"
  done
fi
