#!/bin/bash
# Round 5, final evidence, part 3: c5's profile set; SQ passes for m256 and c4.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r5_profiles
for c in m256 c4; do
  tag=r5; [ $c != m256 ] && tag=r5_$c
  n=80; [ $c = c4 ] && n=40
  bash tools/pmc_sq.sh $tag python3 $R/tools/run_steps.py $c $n > gpurun_out/sq_$tag.log 2>&1
  cp gpurun_out/sq_$tag/summary_sq.json gpurun_out/r5_profiles/${tag}_sq.json 2>/dev/null
  rm -rf gpurun_out/sq_$tag/pass* gpurun_out/sq_$tag/trace
done
bash tools/profile_round.sh r5 c5 2>&1 | tail -12
ls -la gpurun_out/r5_profiles
