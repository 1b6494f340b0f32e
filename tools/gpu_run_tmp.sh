python -m pytest tests/test_train_gpu.py tests/test_grid_rate_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -8 | cut -c1-250
python bench.py --workload train_step --steps 8 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['gsvc_kernel_us_per_step'])"
