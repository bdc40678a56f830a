#!/bin/bash
# Timing breakdown of k_trace with the DIAGNOSTIC library (results are wrong when a switch is set):
# GVOM_TRACE_DEBUG bits: 1 no flush atomics, 2 no tag stores, 4 no endpoint atomics, 8 no step loop, 16 no LDS accumulate
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export GVOM_HIP_LIBRARY=$REPO/g-vom_amd/lib/libgvom_hip_diag.so
for d in 0 1 2 3 4 8 16 17 19 23; do
  GVOM_TRACE_DEBUG=$d python3 $REPO/tools/run_steps.py m256 300 segs=${1:-6} stage | sed "s/^/dbg=$d /"
done
