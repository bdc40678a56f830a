#!/bin/bash
# A/B of k_trace variants on the GPU box: tools/ab_trace.sh "1 4" "m256 c2"
for v in $1; do for c in $2; do
  echo -n "var $v $c: "
  GVOM_TRACE_VARIANT=$v python bench.py --config $c --no-cpu --steps 400 --warmup 50 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value'],1), round(d['ms_per_step']*1e3,1), {k:round(v*1e3,1) for k,v in d['stage_ms'].items()})"
done; done
