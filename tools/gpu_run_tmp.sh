timeout 900 python -m pytest tests/test_train_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -2 | cut -c1-250
python bench.py --workload train_step --steps 30 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k:round(v['avg_us'],1) for k,v in d['kernels'].items() if 'blend' in k})"
