#!/bin/bash
# SQ / TCC counter passes over one command on the GPU box (via gpurun), summarised per kernel into
# gpurun_out/sq_<tag>/summary_sq.json (copy it to profiles/<tag>_sq.json, or re-run tools/summarize_sq.py on the merged passes).  Counters go in their own rocprofv3 runs with --kernel-trace only (never
# with other trace domains).  Usage: tools/pmc_sq.sh <tag> <program> [args...]   (program = python3 ...)
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/sq_$TAG
mkdir -p $OUT $REPO/profiles
export TMPDIR=/tmp
cd /tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH"
 "TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum"
 "GRBM_GUI_ACTIVE"
)
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- "$@" > $OUT/pass$i.log 2> $OUT/pass$i.err || echo "pass $i failed" >&2
  i=$((i+1))
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- "$@" > $OUT/trace.log 2> $OUT/trace.err
python3 $REPO/tools/summarize_sq.py $OUT $OUT/summary_sq.json "$*"
