#!/bin/bash
# Five RD points of the 64-frame 1080p synthetic video (10 000-step fits through the four phases, stream encode -> decode -> evaluate):
#   bash tools/ab/rd_curve.sh   ->  gpurun_out/r05_rd_<lambda>.json / .log
for L in 0.001 0.002 0.004 0.008 0.016; do
  timeout -k 10 400 python tools/fit_synthetic.py --steps 10000 --anchors 100000 --lmbda $L --payload-tol 0.05 --json gpurun_out/r05_rd_$L.json > gpurun_out/r05_rd_$L.log 2>&1
  echo "lambda $L rc $? $(tail -1 gpurun_out/r05_rd_$L.log)"
done
