#!/bin/bash
# Tight binning (default) against the 3-sigma lists (GSVC_RASTER_LOOSE_BINNING=1), early and late in a fit:  bash tools/ab/ab_tight_binning.sh
for streams in 1 2; do
for loose in ${LOOSE:-0 1}; do
  export GSVC_RASTER_STREAMS=$streams
  if [ $loose = 1 ]; then export GSVC_RASTER_LOOSE_BINNING=1; else unset GSVC_RASTER_LOOSE_BINNING; fi
  timeout -k 10 300 python tools/ab/late_stage_profile.py > gpurun_out/r05_s2_tight_${streams}_${loose}.log 2>&1
  echo "== raster streams $streams, loose binning $loose"; grep -v amdgpu.ids gpurun_out/r05_s2_tight_${streams}_${loose}.log | cut -c1-400
done; done
