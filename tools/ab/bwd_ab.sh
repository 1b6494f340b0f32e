# k_blend_bwd_tile per launch at cfg-2 and in the fitting step for the library variants given (names of tools/ab/libgsvc_*.so; "main" = the tree's)
export TMPDIR=/tmp
REPO=$PWD
for v in "$@"; do
  if [ "$v" = main ]; then unset GSVC_LIB_PATH; else export GSVC_LIB_PATH=$REPO/tools/ab/libgsvc_$v.so; fi
  OUT=$REPO/gpurun_out/bwdab_$v; mkdir -p $OUT
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $REPO/bench.py --workload raster_fwdbwd --no-cpu-baseline > $OUT/run.log 2>&1)
  f=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
  echo "variant=$v cfg-2:"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'k_blend' in r['Name'] or 'k_gaussian_bwd' in r['Name']: print('  %-34s calls %4s avg %8.1f us' % (r['Name'][11:45], r['Calls'], float(r['AverageNs'])/1e3))
"
  rm -rf $OUT/raw
  (cd /tmp && GSVC_RASTER_STREAMS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $REPO/bench.py --workload train_step --steps 10 --warmup 2 --pretrain 30 --no-cpu-baseline > $OUT/run2.log 2>&1)
  f=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
  echo "variant=$v fitting step (one stream):"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'k_blend_bwd' in r['Name']: print('  %-34s calls %4s avg %8.1f us' % (r['Name'][11:45], r['Calls'], float(r['AverageNs'])/1e3))
"
  rm -rf $OUT/raw
done
