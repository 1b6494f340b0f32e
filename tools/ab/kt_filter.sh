# per-launch durations of the kernels matching PATTERN in one fitting step (one raster stream): bash tools/ab/kt_filter.sh PATTERN
PAT=$1
REPO=$PWD; OUT=$REPO/gpurun_out/ktf; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
export GSVC_RASTER_STREAMS=1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $REPO/bench.py --workload train_step --steps 6 --warmup 2 --pretrain 30 --no-cpu-baseline > $OUT/run.log 2>&1
t=$(find $OUT/raw -name "*kernel_trace.csv" | head -1)
python3 $REPO/tools/kernel_hist.py $t "$PAT" > $OUT/launches.txt 2>&1
rm -rf $OUT/raw
sed -n '/launches matching/,$p' $OUT/launches.txt
