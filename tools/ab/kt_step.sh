# kernel trace of the train step only (no PMC): bash tools/ab/kt_step.sh <tag>
TAG=${1:-dev}
REPO=$PWD; OUT=$REPO/gpurun_out/kt_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $REPO/bench.py --workload train_step --steps 6 --warmup 2 --pretrain 30 --no-cpu-baseline > $OUT/run.log 2>&1
f=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
python3 $REPO/tools/pmc_extract.py --stats $f 90 > $OUT/kernel_stats.csv
t=$(find $OUT/raw -name "*kernel_trace.csv" | head -1)
python3 $REPO/tools/kernel_hist.py $t > $OUT/launches_per_step.txt 2>&1
rm -rf $OUT/raw
tail -3 $OUT/run.log
