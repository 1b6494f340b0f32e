python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -3 | cut -c1-200
python bench.py > gpurun_out/bench_headline_v5.json 2> gpurun_out/bench_headline_v5.err; tail -c 3000 gpurun_out/bench_headline_v5.json
python bench.py --workload raster_fwdbwd --no-cpu-baseline | tail -1 > gpurun_out/bench_fwdbwd_v5.json
python bench.py --workload train_step --no-cpu-baseline | tail -1 > gpurun_out/bench_train_v6.json
