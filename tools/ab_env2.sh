#!/bin/bash
# as ab_env.sh, printing the whole hipgraph choice
VAR=$1; VALS=$2; shift 2
LEGS="--no-variants --no-unreduced --no-host-leg --no-reference-shape --no-bf16-leg --no-gated-leg --no-cpu-baseline"
for dt in f32 bf16; do
for v in $VALS; do
  env $VAR=$v python3 bench.py --dtype $dt $LEGS --steps 60 --warmup 15 "$@" 2>gpurun_out/ab2.err | python3 -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
j=json.loads(t[-1]) if t else {}
h=j.get('config',{}).get('hipgraph',{})
print('$dt $VAR=$v value %.3f M  ms/step %.3f' % (j.get('value',0)/1e6, j.get('ms_per_step',0)), h.get('critic'), h.get('generator'), h.get('batch_that_trains_both_as_one_graph'), h.get('tuning_ms',{}).get('batch'))
"
done; done
tail -3 gpurun_out/ab2.err | cut -c1-300
