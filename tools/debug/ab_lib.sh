#!/bin/bash
# A/B of two builds of the library on one box: the full bench line of each, alternating.  usage: ab_lib.sh <other .so> [rounds]
cd "$(dirname "$0")/../.." || exit 1
OTHER=$1; R=${2:-2}
for i in $(seq 1 $R); do
  for v in default other; do
    if [ $v = other ]; then export EEGLDM_LIB=$OTHER; else unset EEGLDM_LIB; fi
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parts',{})
print('$v', 'ldm ms', d['ms_per_step'], '| aekl ms', p['aekl_gan_train_step']['ms_per_step'], '| ddim w/s', p['ddim50_sample_decode']['windows_per_s'], 'b1 ms', p['ddim50_sample_decode'].get('batch1_latency_ms'), '| dm ms', p['pixel_dm_train_step']['ms_per_step'])"
  done
done
