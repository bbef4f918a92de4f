#!/bin/bash
# Developer tool (GPU box): one rocprofv3 --pmc pass per argument over tools/pmc_run.py (an argument may hold several counters,
# separated by spaces, that fit one pass).  Counter passes are never combined with other trace domains.
# usage: tools/pmc_collect.sh OUTDIR "CTR_A" "CTR_B CTR_C ..."      (environment, e.g. SYM_ORDER=0, is passed through)
set -e
export TMPDIR=/tmp
R=$PWD
OUT=$1; shift
mkdir -p $OUT
i=0
for pass in "$@"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$OUT/pass$i -- python3 $R/tools/pmc_run.py > $R/$OUT/pass$i.log 2>&1)
  echo "pass$i: $pass -> $(find $OUT/pass$i -name '*counter_collection.csv' | head -1)"
done
