#!/bin/bash
# Timing breakdown of k_trace with the DIAGNOSTIC library (results are wrong when a switch is set):
# GVOM_TRACE_DEBUG bits: 1 no flush atomics, 2 no tag stores, 4 no endpoint atomics, 8 no step loop, 16 no LDS accumulate (the whole
# head branch), 128 no LDS add (the flush then finds nothing to send)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export GVOM_HIP_LIBRARY=$REPO/g-vom_amd/lib/libgvom_hip_diag.so
CFG=${1:-m256}
for d in 0 1 2 8 16 128; do
  GVOM_TRACE_DEBUG=$d python3 $REPO/tools/run_steps.py $CFG ${2:-300} stage | sed "s/^/dbg=$d /"
done
