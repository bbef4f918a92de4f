#!/bin/bash
# Developer tool (GPU box): training-loop cost at the shapes of the reference's own experiment grid (xpert-main.toml:28: six Wilson sets;
# synthetic stand-ins of the same N and D, M = 1024, Matern-3/2 like the experiments, 30 L-BFGS-B iterations).
for spec in "pol 13500 26" "elevators 14939 18" "bike 15642 17" "kin40k 36000 8" "protein 41157 9" "keggundirected 57247 27"; do
  set -- $spec
  echo -n "$1-like: "
  N=$2 D=$3 M=1024 STEPS=30 KERNEL=${KERNEL:-Matern32} python3 tools/train_loop_timing.py 2>&1 | grep -v amdgpu.ids | tail -1
done
