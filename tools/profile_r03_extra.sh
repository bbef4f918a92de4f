#!/bin/bash
# Round-3 evidence beyond tools/profile_round.sh: microbenchmark of the transcendental pipe, per-rank emulation of the library's N-rank loop
# (with its kernel-launch count per iteration from a rocprofv3 kernel trace), rehearsal of bench.py's N > 1 paths.
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r03_extra
hipcc --offload-arch=gfx950 -O3 tools/microbench/trans_overlap.hip -o /tmp/trans_overlap 2>/dev/null && /tmp/trans_overlap > gpurun_out/r03_extra/trans_overlap.log 2>&1
for w in 8 4 2; do
  python3 tools/emulate_rank_native.py $w 30 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03_extra/emulate_rank_native.log
  python3 tools/emulate_rank.py $w 30 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03_extra/emulate_rank_native.log
done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_extra/emul8 -o emul -- python3 $R/tools/emulate_rank_native.py 8 30 > $R/gpurun_out/r03_extra/emul8.log 2>&1)
tools/bench_rehearsal.sh > gpurun_out/r03_extra/bench_rehearsal.log 2>&1
echo extra done
