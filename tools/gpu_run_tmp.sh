python -m pytest tests/test_grid_rate_gpu.py -q -m gpu --tb=short -k ssim 2>&1 | tail -3 | cut -c1-250
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train2 -- python3 $GRAFT_REPO_ROOT/bench.py --workload train_step --steps 5 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_train2.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_train2.log | cut -c1-300
