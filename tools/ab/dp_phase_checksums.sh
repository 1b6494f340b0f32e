set -e
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env GSVC_DIST_BACKEND=gloo GSVC_SHARE_GPU=1 GSVC_DP_PHASE_STEPS=3 GSVC_DP_PHASE_CS=1 "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tests/_dp_phases_worker.py > gpurun_out/r05_s2_ph_$tag.log 2>&1; grep "DP_PHASE FULL" gpurun_out/r05_s2_ph_$tag.log; }
run rowsA GSVC_DP_SPARSE=1
run rowsB GSVC_DP_SPARSE=1
run zownA GSVC_DP_ZOWN=1 GSVC_DP_ZOWN_CHECK=1
run zownB GSVC_DP_ZOWN=1
run rowsNoEarly GSVC_DP_SPARSE=1 GSVC_NO_EARLY_PLAN=1
