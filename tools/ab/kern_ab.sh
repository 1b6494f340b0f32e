# per-launch averages of the kernels matching PATTERN at cfg-2 and in the one-stream fitting step, for library variants:
#   bash tools/ab/kern_ab.sh 'k_preprocess|k_scatter|k_sort' main varA varB      ("main" = the tree's library)
export TMPDIR=/tmp
REPO=$PWD
PAT=$1; shift
for v in "$@"; do
  if [ "$v" = main ]; then unset GSVC_LIB_PATH; else export GSVC_LIB_PATH=$REPO/tools/ab/libgsvc_$v.so; fi
  OUT=$REPO/gpurun_out/kab_$v; mkdir -p $OUT
  for wl in cfg2 step; do
    if [ $wl = cfg2 ]; then ARGS="--workload raster_fwdbwd --no-cpu-baseline"; else ARGS="--workload train_step --steps 10 --warmup 2 --pretrain 30 --no-cpu-baseline"; fi
    (cd /tmp && GSVC_BENCH_NO_500K=1 GSVC_BENCH_NO_4K=1 GSVC_BENCH_NO_PHASES=1 GSVC_RASTER_STREAMS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $REPO/bench.py $ARGS > $OUT/run_$wl.log 2>&1)
    f=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
    echo "variant=$v $wl:"; python3 -c "
import csv,re
for r in csv.DictReader(open('$f')):
    if re.search(r'$PAT', r['Name']): print('  %-40s calls %4s avg %8.1f us  total %8.2f ms' % (r['Name'][6:46], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
"
    rm -rf $OUT/raw
  done
done
