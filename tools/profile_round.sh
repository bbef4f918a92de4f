#!/bin/bash
# Developer tool (GPU box): the rocprofv3 evidence of a round.  usage: tools/profile_round.sh r03
#   1. kernel-trace statistics of the benchmark command itself           -> gpurun_out/<tag>_stats/
#   2. counter passes over tools/pmc_run.py, one --pmc set per pass      -> gpurun_out/<tag>_pmc/   (tools/pmc_collect.sh)
# Counter passes are never combined with other trace domains; the program after `--` is python3 itself (no launcher hop).
set -e
TAG=${1:-r03}
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/${TAG}_stats
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -o bench -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_stats/bench_output.json 2> $R/gpurun_out/${TAG}_stats/bench.err)
echo "stats done: $(find gpurun_out/${TAG}_stats -name '*kernel_stats.csv' | head -1)"
tools/pmc_collect.sh gpurun_out/${TAG}_pmc \
  "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F64" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "GRBM_GUI_ACTIVE"
KIND=matern32 tools/pmc_collect.sh gpurun_out/${TAG}_pmc_m32 \
  "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F64"
echo "profile_round done"
