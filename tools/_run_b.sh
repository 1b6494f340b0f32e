set -x
REPO=$PWD
OUT=$REPO/gpurun_out/s2h
mkdir -p $OUT
python -m pytest tests/test_mlp_gpu.py tests/test_train_gpu.py -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for i in 1 2; do
python bench.py --workload train_step --no-cpu-baseline > $OUT/bench$i.json 2> $OUT/bench.err
python -c "
import json;d=json.loads(open('$OUT/bench$i.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['gsvc_kernel_us_per_step'])"
done
python tools/glue_by_line.py > $OUT/glue.txt 2>&1; grep "aten device" $OUT/glue.txt
python tools/cprofile_step.py > $OUT/cprofile.txt 2>&1; grep "host time" $OUT/cprofile.txt
