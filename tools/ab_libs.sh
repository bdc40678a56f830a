#!/bin/bash
# A/B of library builds on the GPU box: tools/ab_libs.sh <reps> <lib.so> <lib.so> ...  (round-robin, medians)
reps=$1; shift
for r in $(seq $reps); do for l in "$@"; do
  GVOM_HIP_LIBRARY=$PWD/$l python3 bench.py --no-cpu --steps 300 --warmup 30 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$l', round(d['ms_per_step']*1e3,1), ' '.join('%s %.1f' % (k, v*1e3) for k,v in d['stage_ms'].items()))"
done; done | sort | awk '{n[$1]++; t[$1]=t[$1]" "$4; w[$1]=w[$1]" "$2} END {for (k in n) print k, "trace:", t[k], " wall:", w[k]}'
