REPO=$PWD
OUT=$REPO/gpurun_out/s2k
mkdir -p $OUT
python -m pytest tests/test_grid_rate_gpu.py -x -q -k "sampled_rate or rate_kernels" > $OUT/tests0.log 2>&1; tail -15 $OUT/tests0.log
python -m pytest tests/test_train_gpu.py -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for i in 1 2; do
python bench.py --workload train_step --no-cpu-baseline > $OUT/bench$i.json 2> $OUT/bench.err
python -c "
import json;d=json.loads(open('$OUT/bench$i.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['gsvc_kernel_us_per_step'])"
done
python tools/glue_by_line.py > $OUT/glue.txt 2>&1; grep "aten device" $OUT/glue.txt
