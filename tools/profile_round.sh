#!/bin/bash
# rocprofv3 summaries + PMC passes of the bench workloads on the GPU box:  bash tools/profile_round.sh <tag> [workloads...]
# Writes gpurun_out/prof_<tag>/*.csv (copy what is to be judged into profiles/rNN/) and refreshes profiles/pmc_latest.json
# (copied to gpurun_out/ as well: the repo copy on the box does not travel back).  PMC passes are separate runs with
# --kernel-trace only: FETCH_SIZE, WRITE_SIZE (MI355X_MICROARCH.md, rocprofv3 PMC slots) and one SQ pass (VALU issue).
TAG=${1:-dev}; shift
WL=${@:-"raster_fwdbwd train_step"}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for w in $WL; do
  if [ $w = train_step ]; then ARGS="--workload train_step --steps 6 --warmup 2 --pretrain 30 --no-cpu-baseline"; else ARGS="--workload $w --steps 20 --warmup 5 --no-cpu-baseline"; fi
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$w -- python3 $REPO/bench.py $ARGS > $OUT/kt_$w.log 2>&1
  f=$(find $OUT/kt_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 $REPO/tools/pmc_extract.py --stats $f 70 > $OUT/${w}_${TAG}_kernel_stats.csv
  t=$(find $OUT/kt_$w -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && [ $w = train_step ] && python3 $REPO/tools/kernel_hist.py $t > $OUT/${w}_${TAG}_launches_per_step.txt 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${w}_$c -- python3 $REPO/bench.py $ARGS > $OUT/pmc_${w}_$c.log 2>&1
    f=$(find $OUT/pmc_${w}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && grep -E "Kernel_Name|gsvc::" $f > $OUT/${w}_${TAG}_pmc_$c.csv
  done
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/pmc_${w}_SQ -- python3 $REPO/bench.py $ARGS > $OUT/pmc_${w}_SQ.log 2>&1
  f=$(find $OUT/pmc_${w}_SQ -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && grep -E "Kernel_Name|gsvc::k_(blend|preprocess|sort|scatter|gaussian)" $f > $OUT/${w}_${TAG}_pmc_SQ.csv
  if [ $w = train_step ]; then
    # the MLP kernels: matrix-pipe occupancy (what bounds the chain / layer / weight-gradient kernels), with durations from the same pass
    timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_${w}_MFMA -- python3 $REPO/bench.py $ARGS > $OUT/pmc_${w}_MFMA.log 2>&1
    f=$(find $OUT/pmc_${w}_MFMA -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && grep -E "Kernel_Name|gsvc::.*k_(trunk|film|deform|quant_nets|linear)" $f > $OUT/${w}_${TAG}_pmc_MFMA.csv
    [ -n "$f" ] && python3 $REPO/tools/mfma_util.py $OUT/${w}_${TAG}_pmc_MFMA.csv > $OUT/${w}_${TAG}_mfma_utilisation.csv
    rm -rf $OUT/pmc_${w}_MFMA
  fi
  python3 $REPO/tools/pmc_extract.py $w $OUT/${w}_${TAG}_pmc_FETCH_SIZE.csv $OUT/${w}_${TAG}_pmc_WRITE_SIZE.csv $TAG $OUT/${w}_${TAG}_pmc_SQ.csv > $OUT/pmc_extract_$w.log 2>&1
  rm -rf $OUT/kt_$w $OUT/pmc_${w}_FETCH_SIZE $OUT/pmc_${w}_WRITE_SIZE $OUT/pmc_${w}_SQ
done
cp $REPO/profiles/pmc_latest.json $OUT/pmc_latest.json
ls -la $OUT
