#!/bin/bash
# A 400-step fit (gloo stages every collective through the host: ~0.5 s per step) of the synthetic 1080p video on TWO data-parallel ranks sharing one GPU (gloo), once with the replicated row
# exchange and once with z-range ownership (GSVC_DP_ZOWN=1), then encode -> decode -> evaluate on rank 0:  bash tools/ab/dp_fit.sh
for tag in rows zown; do
  if [ $tag = zown ]; then export GSVC_DP_ZOWN=1 GSVC_DP_ZOWN_CHECK=1; else unset GSVC_DP_ZOWN GSVC_DP_ZOWN_CHECK; fi
  GSVC_DIST_BACKEND=gloo GSVC_SHARE_GPU=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
    tools/fit_synthetic.py --steps 400 --anchors 100000 --payload-tol 0.2 --json gpurun_out/r05_dpfit_$tag.json > gpurun_out/r05_dpfit_$tag.log 2>&1
  echo "$tag rc $? $(grep 'RD point' gpurun_out/r05_dpfit_$tag.log | tail -1)"
done
