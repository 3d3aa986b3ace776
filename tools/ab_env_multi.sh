#!/bin/bash
# developer tool: bench.py under several values of one environment knob.  usage: tools/ab_env_multi.sh VAR "v1 v2 v3" -- <bench args>
VAR=$1; VALS=$2; shift 3
for rep in 1 2; do
  for v in $VALS; do
    env $VAR=$v python bench.py "$@" --no-cpu-baseline --whole-fit-maxit 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['ms_per_step'],4), 'syrk', round(d['phases_ms_per_call']['syrk'],4))"
  done
done
