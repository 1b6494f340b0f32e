#!/bin/bash
# k_gaussian_bwd's row summation: per lane only (variant gb100000) against the wave-cooperative form for large rectangles, early and late in a fit,
# one raster stream (exclusive kernel times):  bash tools/ab/build_variant.sh gb100000 raster_bwd.hip -DGSVC_GBWD_COOP_MIN_ROWS=100000; bash tools/ab/ab_gaussian_bwd.sh
export GSVC_RASTER_STREAMS=1
for lib in default gb100000; do
  if [ $lib = default ]; then unset GSVC_LIB_PATH; else export GSVC_LIB_PATH=$PWD/tools/ab/libgsvc_$lib.so; fi
  timeout -k 10 300 python tools/ab/late_stage_profile.py > gpurun_out/r05_s2_late_$lib.log 2>&1
  echo "== $lib"; grep -E "^(early|late)|k_gaussian_bwd" gpurun_out/r05_s2_late_$lib.log | sed -E 's/.*(early|late): iteration ([0-9]+).*: ([0-9.]+ ms\/step).*/\1 \3/; s/.*(k_gaussian_bwd [0-9]+ us).*/   \1/'
done
