#!/bin/bash
# N consecutive `pytest tests -m gpu -x -q` runs in one lease, one line per run (a randomised test that
# passed once is not "green").  Stops at the first failing run.  Usage: tools/repeat_suite.sh <runs> <logfile>
N=${1:-5}; LOG=${2:-gpurun_out/repeat_suite.txt}
: > $LOG
for i in $(seq 1 $N); do
  timeout -k 10 400 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/_suite_run.log 2>&1
  rc=$?
  echo "run $i rc=$rc: $(tail -n 1 gpurun_out/_suite_run.log)" | tee -a $LOG
  if [ $rc -ne 0 ]; then tail -n 40 gpurun_out/_suite_run.log >> $LOG; exit 1; fi
done
