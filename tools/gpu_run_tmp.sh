timeout 900 python -m pytest tests -q -m gpu --tb=short -x 2>&1 | tail -3 | cut -c1-250
python bench.py --workload train_step --steps 20 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['gsvc_kernel_us_per_step'])"
