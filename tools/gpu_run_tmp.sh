python -m pytest tests/test_raster_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -3 | cut -c1-250
python bench.py --workload raster_fwd --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
