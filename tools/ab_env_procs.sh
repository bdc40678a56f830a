#!/bin/bash
# A/B of environment settings across PROCESSES (round-robin; allocation-dependent effects differ from
# process to process): [BENCH_ARGS="--config c3"] tools/ab_env_procs.sh <reps> "ENV=VAL" ...
reps=$1; shift
for r in $(seq $reps); do for e in "$@"; do
  env $e python3 bench.py --no-cpu --steps 300 --warmup 30 $BENCH_ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e'.replace(' ', ','), round(d['ms_per_step']*1e3,1), ' '.join('%s %.1f' % (k, v*1e3) for k,v in d['stage_ms'].items()))"
done; done | awk '{n[$1]++; t[$1]=t[$1]" "$4; e[$1]=e[$1]" "$6; w[$1]=w[$1]" "$2} END {for (k in n) print k, "trace:", t[k], " encode:", e[k], " wall:", w[k]}'
