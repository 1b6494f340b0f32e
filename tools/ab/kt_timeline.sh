# kernel trace of the headline fitting step + its timeline (tools/step_timeline.py): bash tools/ab/kt_timeline.sh <tag> [extra bench args]
TAG=${1:-dev}; shift
REPO=$PWD; OUT=$REPO/gpurun_out/kt_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python3 $REPO/bench.py --workload train_step --steps 6 --warmup 2 --pretrain 30 --no-cpu-baseline "$@" > $OUT/run.log 2>&1
t=$(find $OUT/raw -name "*kernel_trace.csv" | head -1)
python3 $REPO/tools/step_timeline.py $t 2 --list > $OUT/timeline.txt 2>&1
python3 $REPO/tools/step_timeline.py $t 3 > $OUT/timeline_prev.txt 2>&1
rm -rf $OUT/raw
head -60 $OUT/timeline.txt
