#!/bin/bash
# usage: tools/cli_dist_debug.sh TAG DATASET [extra env assignments...]
TAG=$1; DS=$2; shift 2
export PYTHONPATH=$PWD PYTHONFAULTHANDLER=1 CGLB_DIST_BACKEND=gloo CGLB_SHARE_GPU=1
for kv in "$@"; do export "$kv"; done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 -m cglb_amd.cli -b hip -t fp64 -s 3 -l /tmp/two_$TAG train -d $DS -n 8 cglb -k Matern32 -m cglb -i cv -M 24 > gpurun_out/r3/cli_$TAG.log 2>&1
echo "$TAG rc=$?"
grep -c "Segmentation" gpurun_out/r3/cli_$TAG.log
