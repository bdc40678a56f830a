#!/bin/bash
# The randomised campaigns outside the suite, on the final library (one line each into the log).  Usage: tools/campaigns.sh <logfile> [which...]
LOG=${1:-gpurun_out/campaigns.txt}; shift
: > $LOG
run() { echo "== $*" | tee -a $LOG; timeout -k 10 1000 "$@" 2>&1 | tail -n 4 | tee -a $LOG; }
for W in ${@:-many mid shard stats}; do
  case $W in
    many)  run python3 tests/fuzz/fuzz_many.py 1000 3000 ;;
    mid)   run python3 tests/fuzz/fuzz_mid.py 0 300 ;;
    shard) for rep in 1 2 3; do run python3 tests/fuzz/fuzz_shard.py 70000 3000; done ;;
    shard1) run python3 tests/fuzz/fuzz_shard.py 70000 3000 ;;
    stats) run python3 tests/fuzz/fuzz_many.py 5000 300 stats ;;
    p2)    run python3 tests/fuzz/fuzz_many.py 20000 2000 p2; run python3 tests/fuzz/fuzz_many.py 20000 1000 p2 ilv=2 ;;
    dirsort) run python3 tests/fuzz/fuzz_many.py 60000 2000 dirsort; run python3 tests/fuzz/fuzz_many.py 62000 500 dirsort stats ;;
    eager) run python3 tests/fuzz/fuzz_many.py 40000 3000 eager ;;
    ilv)   run python3 tests/fuzz/fuzz_many.py 9000 1500 ilv=2; run python3 tests/fuzz/fuzz_many.py 9000 1500 ilv=4 ;;
  esac
done
