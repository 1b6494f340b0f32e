# per-launch durations of the weight-gradient launches of one fitting step + the products of each launch
TAG=${1:-dev}
REPO=$PWD; OUT=$REPO/gpurun_out/ktw_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
export GSVC_RASTER_STREAMS=1
GSVC_WGRAD_TRACE=1 timeout -k 10 300 python3 $REPO/bench.py --workload train_step --steps 2 --warmup 1 --pretrain 30 --no-cpu-baseline > $OUT/trace.log 2> $OUT/trace.err
grep wgrad_many $OUT/trace.err | tail -14 > $OUT/jobs.txt
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $REPO/bench.py --workload train_step --steps 6 --warmup 2 --pretrain 30 --no-cpu-baseline > $OUT/run.log 2>&1
t=$(find $OUT/raw -name "*kernel_trace.csv" | head -1)
python3 $REPO/tools/kernel_hist.py $t k_linear_wgrad > $OUT/launches.txt 2>&1
rm -rf $OUT/raw
tail -30 $OUT/launches.txt; cat $OUT/jobs.txt
