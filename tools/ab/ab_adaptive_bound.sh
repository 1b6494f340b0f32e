#!/bin/bash
# The measured "GPU or host bound" decision against the row-count heuristic alone (GSVC_NO_ADAPTIVE_BOUND=1), early and late in a 100 k-anchor fit:
#   bash tools/ab/ab_adaptive_bound.sh
for off in 0 1; do
  if [ $off = 1 ]; then export GSVC_NO_ADAPTIVE_BOUND=1; else unset GSVC_NO_ADAPTIVE_BOUND; fi
  timeout -k 10 300 python tools/ab/late_stage_profile.py > gpurun_out/r05_s2_adaptive_$off.log 2>&1
  echo "== GSVC_NO_ADAPTIVE_BOUND=$off"; grep -E "^(early|late)" gpurun_out/r05_s2_adaptive_$off.log | cut -c1-150
done
