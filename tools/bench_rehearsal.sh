#!/bin/bash
# Rehearsal of bench.py's N > 1 code paths on a ONE-GPU box (the driver runs the real N = 2, 4, 8 on a multi-GPU node):
#   a) world 1 with the distributed driver forced and RCCL inside the library (every collective is a real ncclAllReduce / ncclAllGather)
#   b) 2 ranks sharing cuda:0, library loops with collectives by callback over gloo
#   c) 2 ranks sharing cuda:0, host-driven twin over gloo
set -u
mkdir -p gpurun_out/r3
export HSA_ENABLE_IPC_MODE_LEGACY=0
CGLB_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29701 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/r3/bench_force_rccl.json 2> gpurun_out/r3/bench_force_rccl.err
echo "a) rc=$?"; python -c "import json; r=json.load(open('gpurun_out/r3/bench_force_rccl.json')); print(r['driver'], r['driver_note'], r['value'], r['cg_steps'], r['bound'], r['parity_check']['ok'])"
CGLB_BENCH_BACKEND=gloo CGLB_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29702 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/r3/bench_2rank_native.json 2> gpurun_out/r3/bench_2rank_native.err
echo "b) rc=$?"; python -c "import json; r=json.load(open('gpurun_out/r3/bench_2rank_native.json')); print(r['driver'], r['driver_note'], r['value'], r['cg_steps'], r['bound'], r['parity_check']['ok'])"
CGLB_BENCH_DRIVER=python CGLB_BENCH_BACKEND=gloo CGLB_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29703 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/r3/bench_2rank_python.json 2> gpurun_out/r3/bench_2rank_python.err
echo "c) rc=$?"; python -c "import json; r=json.load(open('gpurun_out/r3/bench_2rank_python.json')); print(r['driver'], r['driver_note'], r['value'], r['cg_steps'], r['bound'], r['parity_check']['ok'])"
