# per-kernel averages of the fitting step for library variants: bash tools/ab/step_ab.sh PATTERN variant...
PAT=$1; shift
export TMPDIR=/tmp
REPO=$PWD
for v in "$@"; do
  if [ "$v" = main ]; then unset GSVC_LIB_PATH; else export GSVC_LIB_PATH=$REPO/tools/ab/libgsvc_$v.so; fi
  OUT=$REPO/gpurun_out/stepab_$v; mkdir -p $OUT
  (cd /tmp && GSVC_RASTER_STREAMS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $REPO/bench.py --workload train_step --steps 10 --warmup 2 --pretrain 30 --no-cpu-baseline > $OUT/run.log 2>&1) || { tail -5 $OUT/run.log; exit 1; }
  f=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
  echo "variant=$v:"; python3 -c "
import csv,re
for r in csv.DictReader(open('$f')):
    if re.search('$PAT', r['Name']): print('  %-40s calls %5s avg %8.1f us total %8.2f ms' % (r['Name'][6:46], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
"
  grep -o '"ms_per_step": [0-9.]*' $OUT/run.log | tail -1
  rm -rf $OUT/raw
done
