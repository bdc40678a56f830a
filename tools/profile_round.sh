#!/bin/bash
# Round profile set on the GPU box: for each config the rocprofv3 kernel-trace stats of `bench.py --no-cpu --no-extra` + the
# PMC traffic passes + calibration (tools/profile_gpu.sh), summarised into profiles/<round>[_<config>]_traffic.json and
# profiles/<round>_<config>_kernel_stats.csv.  Usage: tools/profile_round.sh <round tag> [configs...]
R=${1:-r3}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for CFG in ${@:-m256}; do
  TAG=$R; [ "$CFG" != "m256" ] && TAG=${R}_$CFG
  bash $REPO/tools/profile_gpu.sh $TAG --config $CFG > $REPO/gpurun_out/prof_$TAG.log 2>&1
  python3 $REPO/tools/summarize_pmc.py $TAG > $REPO/gpurun_out/prof_${TAG}_summary.txt 2>&1
  f=$(find $REPO/gpurun_out/prof_$TAG/trace -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $REPO/profiles/${R}_${CFG}_kernel_stats.csv && python3 $REPO/tools/lib_identity.py --sidecar $REPO/profiles/${R}_${CFG}_kernel_stats.csv
  cp $REPO/gpurun_out/prof_$TAG/bench_trace.json $REPO/profiles/${R}_bench_under_rocprof_$CFG.json 2>/dev/null
  # what travels back from the GPU box is gpurun_out/ (64 MiB at most): the summaries, not the raw traces
  mkdir -p $REPO/gpurun_out/${R}_profiles
  cp $REPO/profiles/${TAG}_traffic.json $REPO/profiles/${R}_${CFG}_kernel_stats.csv $REPO/profiles/${R}_${CFG}_kernel_stats.csv.meta.json $REPO/profiles/${R}_bench_under_rocprof_$CFG.json \
     $REPO/gpurun_out/prof_${TAG}_summary.txt $REPO/gpurun_out/${R}_profiles/ 2>/dev/null
  rm -rf $REPO/gpurun_out/prof_$TAG
  tail -n 8 $REPO/gpurun_out/${R}_profiles/prof_${TAG}_summary.txt
done
