#!/bin/bash
# Round 5, final evidence, part 2: rocprofv3 kernel stats + PMC traffic passes per config, SQ passes for m256 and c4.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/profile_round.sh r5 "$@" 2>&1 | tail -40
for c in "$@"; do
  [ $c = m256 ] || [ $c = c4 ] || continue
  tag=r5; [ $c != m256 ] && tag=r5_$c
  n=80; [ $c = c4 ] && n=40
  bash tools/pmc_sq.sh $tag python3 $R/tools/run_steps.py $c $n > gpurun_out/sq_$tag.log 2>&1
  cp gpurun_out/sq_$tag/summary_sq.json gpurun_out/r5_profiles/${tag}_sq.json 2>/dev/null
  rm -rf gpurun_out/sq_$tag/pass* gpurun_out/sq_$tag/trace
done
ls -la gpurun_out/r5_profiles
