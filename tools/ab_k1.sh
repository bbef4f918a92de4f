#!/bin/bash
# Developer tool: A/B the K_ff mat-vec of two library builds on the same box (alternating, 3 rounds).
# usage: [N=.. D=.. DTYPE=fp32] tools/ab_k1.sh libA.so libB.so
for i in 1 2 3; do
  for lib in "$@"; do
    echo -n "$(basename $lib): "; CGLB_HIP_LIB=$lib python tools/k1_opts.py kff_variant=2 kff_variant=2 2>&1 | tail -1
  done
done
