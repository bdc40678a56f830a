#!/bin/bash
# quick A/B on the GPU box: tools/ab_quick.sh "ENV=VAL ENV2=VAL" ...  (one bench run per argument)
for e in "$@"; do
  echo -n "$e: "
  env $e python3 bench.py --no-cpu --steps 300 --warmup 30 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step']*1e3,1), {k:round(v*1e3,1) for k,v in d['stage_ms'].items()})"
done
