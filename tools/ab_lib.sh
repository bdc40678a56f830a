#!/bin/bash
# A/B of two builds of the library on one box: the production library against lib/libgvom_hip_var.so (a kernel variant built from
# a modified tree), run in turns by separate processes.  Usage: tools/ab_lib.sh <config> <steps> [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
C=${1:-m256}; N=${2:-300}; ROUNDS=${3:-3}
for i in $(seq 1 $ROUNDS); do
  echo "base: $(python3 $R/tools/run_steps.py $C $N stage)"
  echo "var : $(GVOM_HIP_LIBRARY=$R/g-vom_amd/lib/libgvom_hip_var.so python3 $R/tools/run_steps.py $C $N stage)"
done
