# k_blend_bwd_tile at the configs[3] shape (deep tile lists) for library variants
export TMPDIR=/tmp
REPO=$PWD
for v in "$@"; do
  if [ "$v" = main ]; then unset GSVC_LIB_PATH; else export GSVC_LIB_PATH=$REPO/tools/scratch/libgsvc_$v.so; fi
  OUT=$REPO/gpurun_out/bwdc3_$v; mkdir -p $OUT
  (cd /tmp && GSVC_BENCH_NO_500K=1 GSVC_BENCH_NO_4K=1 GSVC_BENCH_NO_PHASES=1 GSVC_RASTER_STREAMS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $REPO/bench.py --workload train_step --cfg3 --steps 10 --warmup 2 --pretrain 30 --no-cpu-baseline > $OUT/run.log 2>&1)
  f=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
  echo "variant=$v cfg3 step (one stream):"; python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'k_blend' in r['Name']: print('  %-40s calls %4s avg %8.1f us' % (r['Name'][6:46], r['Calls'], float(r['AverageNs'])/1e3))
"
  rm -rf $OUT/raw
done
