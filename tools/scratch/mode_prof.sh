# kernel stats of the fitting step in one phase: bash tools/scratch/mode_prof.sh QUANTIZED
export TMPDIR=/tmp
REPO=$PWD
for PH in "$@"; do
OUT=$REPO/gpurun_out/modeprof_$PH; mkdir -p $OUT
(cd /tmp && GSVC_RASTER_STREAMS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $REPO/tools/scratch/mode_times.py $PH > $OUT/run.log 2>&1)
f=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
python3 $REPO/tools/pmc_extract.py --stats $f 45 > $OUT/kernel_stats.csv
rm -rf $OUT/raw
tail -2 $OUT/run.log
done
