R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01c; mkdir -p $O
python bench.py > $O/bench_headline.json 2> $O/bench_headline.err
python bench.py --workload train_step --no-cpu-baseline | tail -1 > $O/bench_train.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train -- python3 $R/bench.py --workload train_step --steps 8 --warmup 4 --no-cpu-baseline > $O/stats_train.log 2>&1
cd $R; tail -c 1500 $O/bench_headline.json; echo; python -c "
import json; d=json.loads(open('$O/bench_train.json').read()); print(d['ms_per_step'], d['value'], d['roofline'])"
