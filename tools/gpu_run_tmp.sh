python -m pytest tests -q -m gpu -x 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels'])"
python bench.py --steps 20 --warmup 5 --workload raster_fwdbwd --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --workload raster_fwdbwd --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --workload raster_fwdbwd --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc2.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc1/*/ $GRAFT_REPO_ROOT/gpurun_out/pmc2/*/
