#!/bin/bash
# Copy a profile round from gpurun_out/ into profiles/rNN (compact PMC summaries), replacing the previous tag's files:
#   bash tools/publish_profiles.sh r03 v32 v31 <commit>
R=$1; TAG=$2; OLD=$3; COMMIT=$4
P=profiles/$R; G=gpurun_out/prof_$TAG
[ -d $G ] || { echo "no $G"; exit 1; }
git rm -q --cached $P/*_${OLD}_* $P/*_${OLD}.json 2>/dev/null; rm -f $P/*_${OLD}_* $P/*_${OLD}.json
for w in raster_fwdbwd train_step; do
  cp $G/${w}_${TAG}_kernel_stats.csv $P/
  for c in FETCH_SIZE WRITE_SIZE SQ; do python3 tools/pmc_compact.py $G/${w}_${TAG}_pmc_$c.csv > $P/${w}_${TAG}_pmc_${c}_summary.csv; done
done
cp $G/train_step_${TAG}_launches_per_step.txt $P/
cp gpurun_out/bench_headline_$TAG.json $P/bench_headline_$TAG.json
cp gpurun_out/bench_cfg3_$TAG.json $P/bench_train_step_cfg3_$TAG.json
cp gpurun_out/bench_raster_fwd_$TAG.json $P/bench_raster_fwd_$TAG.json
python3 - <<PY
import json
d = json.load(open("$G/pmc_latest.json"))
d["_commit"] = "$COMMIT (library built from this commit; tag $TAG)"
json.dump(d, open("profiles/pmc_latest.json", "w"), indent=1)
PY
sed -i "s/$OLD/$TAG/g" $P/README.md
ls $P
