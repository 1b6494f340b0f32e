#!/bin/bash
# sweep of GSVC_GRID_BWD_WGS for tools/bench_grid_bwd.py under rocprofv3 (kernel durations of k_grid_bwd_lds)
REPO=$PWD; OUT=$REPO/gpurun_out/gridsweep; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for mp in 128 256 384 512; do
  export GSVC_GRID_BWD_WGS=$mp
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$mp -- python3 $REPO/tools/bench_grid_bwd.py > $OUT/log_$mp.txt 2>&1 || exit 1
  f=$(find $OUT/p$mp -name "*kernel_stats.csv" | head -1)
  echo "== target workgroups $mp" >> $OUT/summary.txt
  grep -E "k_grid_bwd_lds|k_grid_absmax" $f | sed "s/(float const.*)\"/\"/" | cut -c1-120 >> $OUT/summary.txt
  grep "call" $OUT/log_$mp.txt >> $OUT/summary.txt
  rm -rf $OUT/p$mp
done
