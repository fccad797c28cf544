#!/bin/bash
# Victim-side A/B of the concurrent-kernel hazard: the narrow gn_bwd_resident blocks beside the LDS-DMA GEMMs of a second
# context, (A) as shipped minus the fence, (B) with every LDS atomic of the kernel replaced by per-thread partials + one writer per sum.
#   bash tools/debug/gn_hazard.sh build     (here, cross-compiling)       bash tools/debug/gn_hazard.sh run   (on the GPU box)
cd "$(dirname "$0")/../.." || exit 1
PKG=synthetic-sleep-eeg-signal-generation-using-latent-diffusion-models_amd
if [ "$1" = build ]; then
  for v in A B C; do
    D="-DEEG_GN_NO_FENCE"; [ $v = B ] && D="$D -DEEG_GN_BWD_NO_LDS_ATOMICS"; [ $v = C ] && D="$D -DEEG_GN_BWD_NO_LDS_ATOMICS -DEEG_GN_HAZ_VERIFY"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Iinclude $D -c $PKG/csrc/norm.hip -o /tmp/norm_$v.o 2>/dev/null || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $PKG/csrc/build/*.o | grep -v /norm.o) /tmp/norm_$v.o -o tools/debug/libeegldm_gn$v.so || exit 1
  done
elif [ "$1" = verify ]; then
  EEGLDM_LIB=tools/debug/libeegldm_gnC.so EEGLDM_GN_BWD_NTH=256 timeout 300 python tools/debug/gn_hazard.py C 256 2>&1 | grep "^hazard\|^HAZ" | sort | uniq -c | sort -rn | head -40
else
  for nth in 256 512 1024; do for v in A B; do
    EEGLDM_LIB=tools/debug/libeegldm_gn$v.so EEGLDM_GN_BWD_NTH=$nth timeout 300 python tools/debug/gn_hazard.py $v $nth 2>&1 | grep "^hazard"
  done; done
fi
