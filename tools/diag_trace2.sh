#!/bin/bash
# As diag_trace.sh, with tuning knobs: tools/diag_trace2.sh <config> <steps> "<dbg values>" key=value ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export GVOM_HIP_LIBRARY=$REPO/g-vom_amd/lib/libgvom_hip_diag.so
CFG=$1; STEPS=$2; DBGS=$3; shift 3
for d in $DBGS; do
  GVOM_TRACE_DEBUG=$d python3 $REPO/tools/run_steps.py $CFG $STEPS stage "$@" | sed "s/^/dbg=$d $* /"
done
