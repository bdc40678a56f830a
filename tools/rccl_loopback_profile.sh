#!/bin/bash
# profiles/<round>_rccl_loopback.txt on the GPU box: the RCCL loopback probe under rocprofv3 (kernel trace), its kernels by stream
# order, and the GPU tests of the same path; with the identity of the library it ran on.  Usage: tools/rccl_loopback_profile.sh <round tag>
R=${1:-r6}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${R}_rccl_loopback.txt
export TMPDIR=/tmp GVOM_COMM_TIMEOUT_S=120
cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_loop -o loop -- python3 $REPO/tools/rccl_loopback_probe.py 2,4 6 > $REPO/gpurun_out/loop_prof_run.txt 2>&1
cd $REPO
timeout -k 10 900 python3 -m pytest tests/test_hip_sharded.py -x -q -m gpu -k "moving_window or statistics or full_size or one_rank or gives_up" > gpurun_out/loop_tests.txt 2>&1
{
  echo "# RCCL loopback on a one-GPU MI355X box (round ${R#r}).  Commands (tools/rccl_loopback_profile.sh $R, via gpurun):"
  echo "#   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_loop -o loop -- python3 tools/rccl_loopback_probe.py 2,4 6"
  echo "#   python3 tools/rocpd_summary.py gpurun_out/prof_loop/loop_results.db --stream-excerpt 14"
  echo "# library:"; python3 tools/lib_identity.py | sed 's/^/#   /'
  echo; echo "## probe output"; grep -v "simple_timer\|^W20\|^E20\|^I20" gpurun_out/loop_prof_run.txt
  echo; echo "## kernels (rocprofv3 kernel trace)"; python3 tools/rocpd_summary.py gpurun_out/prof_loop/loop_results.db --stream-excerpt 14
  echo; echo "## GPU tests of the same path (tests/test_hip_sharded.py -k 'moving_window or statistics or full_size or one_rank or gives_up')"; cat gpurun_out/loop_tests.txt
} > $OUT
rm -rf gpurun_out/prof_loop
tail -5 $OUT
