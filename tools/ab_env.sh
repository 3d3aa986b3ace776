#!/bin/bash
# developer tool: bench.py under two values of one environment knob.  usage: tools/ab_env.sh VAR v1 v2 -- <bench args>
VAR=$1; A=$2; B=$3; shift 4
for rep in 1 2; do
  for v in $A $B; do
    env $VAR=$v python bench.py "$@" --no-cpu-baseline --whole-fit-maxit 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['phases_ms_per_call'].items() if x})"
  done
done
