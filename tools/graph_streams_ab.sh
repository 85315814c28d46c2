#!/bin/bash
# A/B on the GPU box: whole-step hipGraph replay at the headline geometry, captured on one stream or with the side streams' fork / join.
LEGS="--no-variants --no-unreduced --no-host-leg --no-reference-shape --no-bf16-leg --no-gated-leg --no-cpu-baseline"
for dt in f32 bf16; do
for gs in 0 1; do
  PTTS_GRAPH_STREAMS=$gs timeout -k 10 240 python3 bench.py --graph --dtype $dt $LEGS --steps 36 --warmup 12 2>gpurun_out/gs_$dt_$gs.err | python3 -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
j=json.loads(t[-1]) if t else {}
print('$dt graph streams=$gs value %.3f M  ms/step %.3f  critic %.3f  gen %.3f' % (j.get('value',0)/1e6, j.get('ms_per_step',0), j.get('critic_step_ms',0), j.get('generator_step_ms',0)))
"
done; done
