#!/bin/bash
# Round-6 developer tool: A/B of the AEKL / GAN step between two BUILDS of the library on one box, alternating processes.
#   bash tools/r06/aekl_ab_lib.sh tools/r06/libeegldm_<variant>.so [rounds]
cd "$(dirname "$0")/../.." || exit 1
OTHER=$1; R=${2:-3}
for i in $(seq 1 $R); do
  unset EEGLDM_LIB; echo -n "default : "; python tools/r06/aekl_ab.py 2>&1 | grep "^default"
  export EEGLDM_LIB=$OTHER; echo -n "other   : "; python tools/r06/aekl_ab.py 2>&1 | grep "^default"
done
