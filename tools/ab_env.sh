#!/bin/bash
# A/B on the GPU box: bench.py headline loop (side legs off) for each value of one environment switch.
#   bash tools/ab_env.sh PTTS_SIDE_BWD_FIRST "0 1" [extra bench args]
VAR=$1; VALS=$2; shift 2
LEGS="--no-variants --no-unreduced --no-host-leg --no-reference-shape --no-bf16-leg --no-gated-leg --no-cpu-baseline"
for dt in f32 bf16; do
for v in $VALS; do
  env $VAR=$v python3 bench.py --dtype $dt $LEGS --steps 60 --warmup 15 "$@" 2>/dev/null | python3 -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
j=json.loads(t[-1]) if t else {}
print('$dt $VAR=$v value %.3f M  ms/step %.3f  critic %.3f  gen %.3f' % (j.get('value',0)/1e6, j.get('ms_per_step',0), j.get('critic_step_ms',0), j.get('generator_step_ms',0)), j.get('config',{}).get('hipgraph',{}).get('critic'), j.get('config',{}).get('hipgraph',{}).get('generator'))
"
done; done
