#!/bin/bash
# A/B the LDM step under environment variants: tools/debug/ab.sh "VAR=1" "VAR=2 OTHER=x" ...   (an empty string = defaults)
for cfg in "$@"; do
  out=$(env $cfg python bench.py --no-parts --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["config"]["final_loss"])')
  echo "AB [$cfg] $out"
done
