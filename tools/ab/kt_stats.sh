REPO=$PWD; OUT=$REPO/gpurun_out/prof_r05b; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/bench.py --workload train_step --steps 6 --warmup 2 --pretrain 30 --no-cpu-baseline > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
python3 $REPO/tools/pmc_extract.py --stats $f 70 > $OUT/train_step_r05b_kernel_stats.csv
t=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python3 $REPO/tools/kernel_hist.py $t > $OUT/train_step_r05b_launches_per_step.txt 2>&1
python3 $REPO/tools/step_timeline.py $t 2 > $OUT/train_step_r05b_timeline.txt 2>&1
rm -rf $OUT/kt
head -14 $OUT/train_step_r05b_kernel_stats.csv | cut -c1-120
